"""The generated constant tables (tools/gen_tables.py -> tables_imdct.inc)."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _parse_inc():
    txt = open(os.path.join(ROOT, "libxaac_amd", "csrc", "tables_imdct.inc")).read()
    out = {}
    for m in re.finditer(r"xaac_tab_(\w+)\[(\d+)\] = \{([^}]*)\}", txt):
        vals = np.array([int(v) for v in m.group(3).replace("\n", " ").split(",") if v.strip()], np.int64)
        assert len(vals) == int(m.group(2))
        out[m.group(1)] = vals
    return out


def test_committed_include_matches_generator():
    import gen_tables
    t = gen_tables.tables()
    inc = _parse_inc()
    for name in ("pre_cs", "win_long_sine", "win_long_kbd", "win_short_sine", "win_short_kbd", "digrev_long"):
        assert np.array_equal(inc[name], t[name]), name
    assert np.array_equal(inc["fft_tw"] >> 16, t["fft_tw_hi"])


def test_window_power_complementarity():
    """Princen-Bradley: w[i]^2 + w[N-1-i]^2 == 1 (to table precision) for every window"""
    inc = _parse_inc()
    for name in ("win_long_sine", "win_long_kbd", "win_short_sine", "win_short_kbd"):
        w = inc[name].astype(np.float64) / 32768.0
        s = w[0::2] ** 2 + w[1::2] ** 2
        assert np.max(np.abs(s - 1.0)) < 1e-4, name


def test_tables_equal_reference_rom():
    import derive_table_fixups as d
    so = os.path.join(ROOT, "oracle", "_ref", "libxaacdec_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built")
    import gen_tables
    ref = d.reference_tables(so)
    mine = gen_tables.tables()
    for name, want in ref.items():
        assert np.array_equal(mine[name], want), name


def _regenerated_equals_committed(tool, out_name, tmp_path, where="csrc"):
    """tools/gen_tables_{qmf,sbr,ps}.py read the constants out of the compiled reference's ROM image; the committed
    .inc must be exactly what they produce today"""
    import importlib
    import os
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libxaacdec_ref.so")
    if not os.path.exists(ref_so):
        pytest.skip("oracle/_ref/libxaacdec_ref.so missing (built where /root/reference exists)")
    mod = importlib.import_module(tool)
    out = str(tmp_path / out_name)
    mod.emit(mod.reference_tables(), out)
    committed = os.path.join(ROOT, "libxaac_amd", where, out_name)
    assert open(out).read() == open(committed).read(), "%s is stale: rerun tools/%s.py" % (out_name, tool)


def test_qmf_tables_equal_reference_rom(tmp_path):
    _regenerated_equals_committed("gen_tables_qmf", "tables_qmf.inc", tmp_path)


def test_sbr_tables_equal_reference_rom(tmp_path):
    _regenerated_equals_committed("gen_tables_sbr", "tables_sbr.inc", tmp_path)


def test_ps_tables_equal_reference_rom(tmp_path):
    _regenerated_equals_committed("gen_tables_ps", "tables_ps.inc", tmp_path)


def test_esbr_qmf_tables_equal_reference_rom(tmp_path):
    _regenerated_equals_committed("gen_tables_qmf_esbr", "tables_qmf_esbr.inc", tmp_path)


def test_usac_tables_equal_reference_rom(tmp_path):
    _regenerated_equals_committed("gen_tables_usac", "tables_usac.inc", tmp_path)


def test_esbr_float_tables_equal_reference_rom(tmp_path):
    _regenerated_equals_committed("gen_tables_esbr", "tables_esbr.inc", tmp_path)


def test_eld_qmf_tables_equal_reference_rom(tmp_path):
    _regenerated_equals_committed("gen_tables_qmf_eld", "tables_qmf_eld.inc", tmp_path)


def test_imdct960_tables_equal_reference_rom(tmp_path):
    _regenerated_equals_committed("gen_tables_imdct960", "tables_imdct960.inc", tmp_path)


def test_imdct_ld_tables_equal_reference_rom(tmp_path):
    _regenerated_equals_committed("gen_tables_imdct_ld", "tables_imdct_ld.inc", tmp_path)


def test_hbe_tables_equal_reference_rom(tmp_path):
    _regenerated_equals_committed("gen_tables_hbe", "tables_hbe.inc", tmp_path)


def test_hbe_dft_host_tables_equal_reference_rom(tmp_path):
    _regenerated_equals_committed("gen_tables_hbe_dft_host", "tables_hbe_dft.inc", tmp_path, where="host")


def test_pvc_tables_equal_reference_rom(tmp_path):
    _regenerated_equals_committed("gen_tables_pvc", "tables_pvc.inc", tmp_path)


def test_esbr_ps_tables_equal_reference_rom_and_libm(tmp_path):
    """ROM members as exact float literals + the mixing-matrix table re-derived with this machine's C library"""
    _regenerated_equals_committed("gen_tables_esbr_ps", "tables_esbr_ps.inc", tmp_path)


def test_esbr_tables_are_the_q31_versions_of_the_q15_ones():
    """independent of the ROM: the 32-bit eSBR constants are the 16-bit bank constants at 16 more fractional bits
    (prototype filter, radix-4 twiddles, modulation twiddles), to rounding"""
    def parse(path, prefix):
        txt = open(os.path.join(ROOT, "libxaac_amd", "csrc", path)).read()
        return {m.group(1): np.array([int(v) for v in m.group(3).replace("\n", " ").split(",") if v.strip()], np.int64)
                for m in re.finditer(r"%s(\w+)\[(\d+)\] = \{([^}]*)\}" % prefix, txt)}
    q15, q31 = parse("tables_qmf.inc", "xaac_qmf_"), parse("tables_qmf_esbr.inc", "xaac_qmf_esbr_")
    for a, b in (("qmf_c", "qmf_c"), ("w_32", "w_32"), ("w_16", "w_16"), ("sin_cos_twiddle_l64", "sin_cos_twiddle_l64"),
                 ("alt_sin_twiddle_l64", "alt_sin_twiddle_l64"), ("sin_cos_twiddle_l32", "sin_cos_twiddle_l32"),
                 ("alt_sin_twiddle_l32", "alt_sin_twiddle_l32"), ("t_cos_sin_l32", "t_cos_sin_l32")):
        assert np.max(np.abs(q31[b] / 65536.0 - q15[a])) <= 1.0, a


def test_aac_syntax_tables_equal_reference_rom(tmp_path):
    """the host parser's code books (every code word listed by probing the reference's own lookup), inverse quantiser, gains,
    TNS and band tables"""
    _regenerated_equals_committed("gen_tables_aac", "tables_aac.inc", tmp_path, where="host")


def test_sbr_side_info_tables_equal_reference_rom(tmp_path):
    """the host parser's SBR / PS code books, FIXFIX grids and log2 table"""
    _regenerated_equals_committed("gen_tables_sbr_side", "tables_sbr_side.inc", tmp_path, where="host")


def test_ps_rotation_factors_hold_no_minus_one():
    """sbr_ps_frame.h's packed all-pass (xp_allpass_packed) negates the imaginary parts of the fractional-delay phase factors
    into shorts and relies on two sample x factor products never summing to 2^31: true as long as no factor is -32768"""
    import re
    src = open(os.path.join(ROOT, "libxaac_amd", "csrc", "tables_ps.inc")).read()
    for name in ("frac_delay_phase_fac_qmf_re_im", "frac_delay_phase_fac_qmf_sub_re_im", "frac_delay_phase_fac_qmf_ser_re_im",
                 "frac_delay_phase_fac_qmf_sub_ser_re_im"):
        body = re.search(r"/\* %s \*/ \{(.*?)\}" % name, src, re.S).group(1)
        vals = [int(v) for v in re.findall(r"-?\d+", body)]
        assert len(vals) >= 32 and min(vals) > -32768 and max(vals) <= 32767, name
