"""Float HF generator + envelope adjuster of the reference's default SBR path (Path A): the oracle (oracle/oracle_esbr.cpp,
arithmetic of libxaac_amd/csrc/esbr_core.h) against the compiled reference's own ixheaacd_generate_hf
(sbrdec_lpfuncs.c:981) and ixheaacd_sbr_env_calc (esbr_envcal.c:71), driven through oracle/ref_esbr_adapter.c.  Side info:
the headers / grids / inverse-filter modes / harmonics of the captured HE-AAC streams (tests/golden/sbr_lp_records),
with random float envelope and noise-floor data over a wide range, all limiter settings, interpolation / smoothing on
and off, master tables with and without a cross-over offset, resets, inter-TES shaping.  Compared as raw words: every
float of both output matrices and the whole persistent state (gain / noise histories, limiter tables, patch borders,
chirp factors, harmonic flags, phase indices)."""
import ctypes
import os

import numpy as np
import pytest

import sbr_capture as c
from esbr_structs import EsbrSide, EsbrState, new_state

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PF = ctypes.POINTER(ctypes.c_float)


def bind(lib, name):
    fn = getattr(lib, name)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [PF] * 4
    return fn


def make_side(rng, h, f, prev_modes, frame_no, xover_extra, tes):
    sd = EsbrSide()
    sd.out_sampling_freq = int(rng.choice([32000, 44100, 48000]))
    sd.limiter_bands = int(rng.integers(0, 4))
    hi = [h.freq_band_tbl_hi[i] for i in range(h.num_sf_bands[1] + 1)]
    fm = ([hi[0] - 2 * (xover_extra - j) for j in range(xover_extra)] if hi[0] - 2 * xover_extra >= 4 else []) + hi
    sd.num_mf_bands = len(fm) - 1
    for i, v in enumerate(fm):
        sd.f_master_tbl[i] = v
    sd.qmf_sb_prev = h.sub_band_start
    sd.reset_flag = 1 if frame_no == 0 or rng.integers(0, 9) == 0 else 0
    for i in range(10):
        sd.sbr_invf_mode_prev[i] = prev_modes[i]
    if tes:
        for i in range(8):
            sd.inter_temp_shape_mode[i] = int(rng.integers(0, 4))
    env = (2.0 ** rng.uniform(-4, 40, 448)).astype(np.float32)
    nf = (2.0 ** rng.uniform(-14, 6, 10)).astype(np.float32)
    for i in range(448):
        sd.flt_env_sf_arr[i] = env[i]
    for i in range(10):
        sd.flt_noise_floor[i] = nf[i]
    return sd


def qmf_matrices(rng, level):
    q = np.zeros((2, 72, 64), np.float32)
    q[:, :, :32] = (rng.standard_normal((2, 72, 32)) * level).astype(np.float32)
    tone = rng.integers(1, 30)
    q[0, :, tone] += np.float32(level * 8) * np.cos(np.arange(72) * 0.7).astype(np.float32)
    return np.ascontiguousarray(q[0]), np.ascontiguousarray(q[1])


def state_words(st):
    return np.frombuffer(bytes(st), np.uint32)


def bind_h(lib, name):
    fn = getattr(lib, name)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [PF] * 6 + [ctypes.POINTER(ctypes.c_int32)]
    return fn


def run_chain(oracle, reference, recs, seed, xover_extra=0, tes=False, n_frames=40, harmonic=False):
    """harmonic: the HF generator's harmonic branch (sbr_patching_mode 0: the input is the transposer's rows, here random
    ones) on most frames, LPP patching on the others, so that the limiter bands are remade at every change"""
    if harmonic:
        ref_h, orc_h = bind_h(reference.lib, "ref_esbr_hf_env_h"), bind_h(oracle.lib, "xo_esbr_hf_env_h")
    ref_fn, orc_fn = bind(reference.lib, "ref_esbr_hf_env"), bind(oracle.lib, "xo_esbr_hf_env")
    rng = np.random.default_rng(seed)
    st_r, st_o = new_state(), new_state()
    prev_modes = [0] * 10
    prev_tables = None
    start = int(rng.integers(0, max(1, len(recs) - n_frames)))
    done = flattened = 0
    for n, rec in enumerate(recs[start:start + n_frames]):
        h, f = c.Header.from_buffer_copy(bytes(rec["header"])), c.Frame.from_buffer_copy(bytes(rec["frame"]))
        if not f.apply_processing:
            continue
        h.interpol_freq = int(rng.integers(0, 2))
        h.smoothing_mode = int(rng.integers(0, 2))
        h.limiter_gains = int(rng.integers(0, 4))
        for i in range(h.num_nf_bands):
            f.sbr_invf_mode[i] = int(rng.integers(0, 4))
        if rng.integers(0, 3) == 0:
            for i in range(h.num_sf_bands[1]):
                f.add_harmonics[i] = int(rng.integers(0, 5) == 0)
        sd = make_side(rng, h, f, prev_modes, n, xover_extra, tes)
        tables = bytes(h)[12:]                        # a new header comes with a reset (the parser raises reset_flag)
        if prev_tables != tables:
            sd.reset_flag = 1
        prev_tables = tables
        qre, qim = qmf_matrices(rng, float(2.0 ** rng.integers(0, 16)))
        outs = []
        if harmonic:
            sd.harmonic_sbr = int(rng.integers(0, 4) != 0)
            sd.pitch_in_bins = int(rng.integers(0, 128))
            lvl = float(2.0 ** rng.integers(0, 16))
            ph = [(rng.standard_normal((40, 64)) * lvl).astype(np.float32) for _ in range(2)]
            ph[0][:, 40] += np.float32(lvl * 6) * np.cos(np.arange(40) * 1.1).astype(np.float32)
            sb = h.sub_band_start
            xo = (ctypes.c_int32 * 6)(sb, min(64, 2 * sb), min(64, 3 * sb) if rng.integers(0, 2) else 0, 0, 0, 0)
        if rng.integers(0, 3) == 0:                   # the ENHSBR element's pre-flattening flag (acts on LPP patches)
            sd.harmonic_sbr |= 2
            flattened += int(not (sd.harmonic_sbr & 1) and f.apply_processing != 0)
        for fn, st in (((ref_h if harmonic else ref_fn), st_r), ((orc_h if harmonic else orc_fn), st_o)):
            ore, oim = np.zeros((72, 64), np.float32), np.zeros((72, 64), np.float32)
            a, b = qre.copy(), qim.copy()
            if harmonic:
                pa, pb = ph[0].copy(), ph[1].copy()
                rc = fn(ctypes.byref(h), ctypes.byref(f), ctypes.byref(sd), ctypes.byref(st), a.ctypes.data_as(PF),
                        b.ctypes.data_as(PF), ore.ctypes.data_as(PF), oim.ctypes.data_as(PF), pa.ctypes.data_as(PF),
                        pb.ctypes.data_as(PF), xo)
            else:
                rc = fn(ctypes.byref(h), ctypes.byref(f), ctypes.byref(sd), ctypes.byref(st), a.ctypes.data_as(PF),
                        b.ctypes.data_as(PF), ore.ctypes.data_as(PF), oim.ctypes.data_as(PF))
            outs.append((rc, ore, oim, a, b))
        (rc_r, ore_r, oim_r, a_r, b_r), (rc_o, ore_o, oim_o, a_o, b_o) = outs
        assert rc_r == rc_o, (n, rc_r, rc_o)
        if rc_r:
            st_r, st_o = new_state(), new_state()
            continue
        bad = np.argwhere(ore_r.view(np.uint32) != ore_o.view(np.uint32))
        assert bad.size == 0, (n, len(bad), bad[:4].tolist(), ore_r[tuple(bad[0])], ore_o[tuple(bad[0])])
        assert np.array_equal(oim_r.view(np.uint32), oim_o.view(np.uint32)), n
        assert np.array_equal(a_r.view(np.uint32), a_o.view(np.uint32)) and np.array_equal(b_r.view(np.uint32), b_o.view(np.uint32))
        if not np.array_equal(state_words(st_r), state_words(st_o)):
            d = [x for x in c.diff_state(st_r, st_o) if x[0] not in ("ana", "syn")]
            raise AssertionError((n, d))
        assert np.any(ore_r[2:34, h.sub_band_start:h.sub_band_end] != 0)
        prev_modes = [f.sbr_invf_mode[i] for i in range(10)]
        done += 1
    assert done >= n_frames // 2 and (flattened > 0 or harmonic)


@pytest.fixture(scope="module")
def recs():
    return c.read_records(os.path.join(ROOT, "tests", "golden", "sbr_lp_records.bin.gz")) + \
        [r for r in c.read_records(os.path.join(ROOT, "tests", "golden", "sbr_hq_ps_records.bin.gz"))]


@pytest.mark.parametrize("seed", range(6))
def test_hf_env_chain(oracle, reference, recs, seed):
    run_chain(oracle, reference, recs, 100 + seed)


@pytest.mark.parametrize("seed", range(3))
def test_hf_env_chain_with_xover_offset(oracle, reference, recs, seed):
    run_chain(oracle, reference, recs, 200 + seed, xover_extra=1 + seed)


@pytest.mark.parametrize("seed", range(5))
def test_hf_env_chain_with_harmonic_patching(oracle, reference, recs, seed):
    run_chain(oracle, reference, recs, 400 + seed, harmonic=True, tes=seed == 4)


@pytest.mark.parametrize("seed", range(3))
def test_hf_env_chain_with_inter_tes(oracle, reference, recs, seed):
    run_chain(oracle, reference, recs, 300 + seed, tes=True)
