"""Pins the HQ-SBR + parametric-stereo oracle (HE-AACv2, xo_sbr_dec_hq) to the compiled reference on (a) every
ixheaacd_sbr_dec call captured while the reference decodes freshly encoded HE-AACv2 streams and (b) the same
frames with fuzzed SBR and PS side info pushed through the reference by oracle/ref_sbr_adapter.c, state chained
through the reference.  Needs oracle/_ref."""
import ctypes
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

import sbr_capture as cap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P16 = ctypes.POINTER(ctypes.c_int16)


@pytest.fixture(scope="module")
def captures(reference, tmp_path_factory):
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "xaacdec_capture")):
        pytest.skip("capture build of the reference decoder missing")
    d = tmp_path_factory.mktemp("streams_v2")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_test_streams.py"), str(d), "29"],
                          stdout=subprocess.DEVNULL)
    files = sorted(glob.glob(os.path.join(str(d), "*aot29*.cap")))
    assert len(files) >= 6
    return files


def _run(lib, fn, h, f, st, pf, ps, pin):
    out = np.zeros(4096, np.int16)
    rc = getattr(lib, fn)(ctypes.byref(h), ctypes.byref(f), ctypes.byref(st), ctypes.byref(pf), ctypes.byref(ps),
                          pin.ctypes.data_as(P16), 1, out.ctypes.data_as(P16), 2)
    return rc, out


def test_boundary_struct_sizes():
    assert ctypes.sizeof(cap.PsFrame) == 972 and ctypes.sizeof(cap.PsState) == 7764


def test_captured_calls_bit_exact(oracle, reference, captures):
    n = 0
    for path in captures:
        for r in cap.read_records(path):
            assert r["low_pow"] == 0 and r["ps"] == 1
            st = cap.State.from_buffer_copy(bytes(r["st0"]))
            ps = cap.PsState.from_buffer_copy(bytes(r["ps0"]))
            rc, out = _run(oracle.lib, "xo_sbr_dec_hq", r["header"], r["frame"], st, r["ps_frame"], ps,
                           np.ascontiguousarray(r["pcm_in"]))
            assert rc == r["ret"], (path, r["call"])
            assert np.array_equal(out[0::2], r["pcm_out"][0]) and np.array_equal(out[1::2], r["pcm_out"][1]), (path, r["call"])
            assert not cap.diff_state(st, r["st1"]), (path, r["call"], cap.diff_state(st, r["st1"])[:3])
            assert not cap.diff_state(ps, r["ps1"]), (path, r["call"], cap.diff_state(ps, r["ps1"])[:3])
            n += 1
    assert n > 500


def test_fuzzed_side_info_chained_through_reference(oracle, reference, captures):
    """What freshly encoded streams never exercise: inverse-filtering modes (the complex LPC filter), gain
    smoothing, the other limiter gains, energies per scale-factor band, the fine IID quantiser, several PS
    envelopes per frame, sinusoids."""
    rng = np.random.default_rng(11)
    n = 0
    for path in captures[::2]:
        st_r = ps_r = None
        for r in cap.read_records(path):
            h = cap.Header.from_buffer_copy(bytes(r["header"]))
            f = cap.Frame.from_buffer_copy(bytes(r["frame"]))
            pf = cap.PsFrame.from_buffer_copy(bytes(r["ps_frame"]))
            for i in range(h.num_if_bands):
                f.sbr_invf_mode[i] = int(rng.integers(0, 4))
            h.limiter_gains = int(rng.integers(0, 4))
            h.interpol_freq = int(rng.integers(0, 2))
            h.smoothing_mode = int(rng.integers(0, 2))
            if rng.integers(0, 4) == 0:
                for i in range(h.num_sf_bands[1]):
                    f.add_harmonics[i] = int(rng.integers(0, 3) == 0)
            pf.iid_quant = int(rng.integers(0, 2))
            nenv = int(rng.integers(1, 5))
            borders = [0] + sorted(rng.choice(np.arange(1, 32), nenv - 1, replace=False).tolist()) + [32]
            for e in range(7):
                pf.border_position[e] = borders[e] if e < len(borders) else 0
            lim = 15 if pf.iid_quant else 7
            for e in range(nenv):
                for b in range(20):
                    pf.iid_par_table[e][b] = int(rng.integers(-lim, lim + 1))
                    pf.icc_par_table[e][b] = int(rng.integers(0, 8))
            if st_r is None:
                st_r = cap.State.from_buffer_copy(bytes(r["st0"]))
                ps_r = cap.PsState.from_buffer_copy(bytes(r["ps0"]))
            st_o = cap.State.from_buffer_copy(bytes(st_r))
            ps_o = cap.PsState.from_buffer_copy(bytes(ps_r))
            pin = np.ascontiguousarray(r["pcm_in"])
            ra, oa = _run(reference.lib, "ref_sbr_dec_hq", h, f, st_r, pf, ps_r, pin)
            rb, ob = _run(oracle.lib, "xo_sbr_dec_hq", h, f, st_o, pf, ps_o, pin)
            assert ra == rb and np.array_equal(oa, ob), (path, r["call"], int(np.sum(oa != ob)))
            assert not cap.diff_state(st_o, st_r), (path, r["call"], cap.diff_state(st_o, st_r)[:3])
            assert not cap.diff_state(ps_o, ps_r), (path, r["call"], cap.diff_state(ps_o, ps_r)[:3])
            n += 1
    assert n > 250
