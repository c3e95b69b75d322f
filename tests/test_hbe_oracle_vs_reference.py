"""The harmonic transposer's polyphase banks: the oracle (oracle/oracle_hbe.cpp, arithmetic of
libxaac_amd/csrc/hbe_poly.h) against the compiled reference's own ixheaacd_real_synth_filt / ixheaacd_complex_anal_filt
(decoder/ixheaacd_esbr_polyphase.c:157 / :48, driven by oracle/ref_hbe_adapter.c on a transposer the reference
initialises): every float of the state identical, bit for bit, over chains of frames with the delay lines carried, for
every bank size (start bands 0 .. 32 give synth_size 4, 8, 12, 16, 20)."""
import ctypes

import numpy as np
import pytest

from hbe_structs import HbeState, new_state

PF = ctypes.POINTER(ctypes.c_float)


def _p(a):
    return a.ctypes.data_as(PF)


def bits(st):
    return np.frombuffer(bytes(st), np.uint32)


def qmf_columns(rng, kind):
    if kind == 0:
        a = 2.0 ** rng.integers(-10, 18)
        return [(rng.standard_normal((32, 64)) * a).astype(np.float32) for _ in range(2)]
    if kind == 1:  # a few tones
        re, im = np.zeros((32, 64), np.float32), np.zeros((32, 64), np.float32)
        for _ in range(4):
            k, ph, w = int(rng.integers(0, 40)), rng.uniform(0, 6.28), rng.uniform(0, 3.0)
            re[:, k] += (3000 * np.cos(ph + w * np.arange(32))).astype(np.float32)
            im[:, k] += (3000 * np.sin(ph + w * np.arange(32))).astype(np.float32)
        return [re, im]
    if kind == 2:
        return [np.zeros((32, 64), np.float32) for _ in range(2)]
    x = [(rng.standard_normal((32, 64)) * 1e-3).astype(np.float32) for _ in range(2)]
    x[0][int(rng.integers(0, 32)), int(rng.integers(0, 24))] = 3.0e38  # overflow to inf / nan downstream
    return x


@pytest.mark.parametrize("start_band", [0, 3, 4, 7, 11, 12, 15, 19, 20, 23, 27, 28, 32])
def test_banks_chain(oracle, reference, start_band):
    rng = np.random.default_rng(500 + start_band)
    o_syn, o_ana = oracle.lib.xo_hbe_real_synth, oracle.lib.xo_hbe_cplx_anal
    r_syn, r_ana = reference.lib.ref_hbe_real_synth, reference.lib.ref_hbe_cplx_anal
    for fn in (o_syn, r_syn):
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(HbeState), PF, PF, ctypes.c_int]
    for fn in (o_ana, r_ana):
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(HbeState)]
    so, sr = new_state(start_band), new_state(start_band)
    s = so.synth_size
    for frame in range(7):
        kind = 0 if frame < 3 else int(rng.integers(0, 4)) if frame < 6 else 3
        re, im = qmf_columns(rng, kind)
        for st in (so, sr):  # ixheaacd_qmf_hbe_apply's shift of the time signal (hbe_trans.c:235-238)
            buf = np.frombuffer(st, np.float32, 1088, HbeState.input_buf.offset)
            buf[:s] = buf[32 * s:33 * s].copy()
        assert o_syn(ctypes.byref(so), _p(re), _p(im), 32) == 0
        assert r_syn(ctypes.byref(sr), _p(re), _p(im), 32) == 0
        assert np.array_equal(bits(so), bits(sr)), "synthesis bank, frame %d" % frame
        assert o_ana(ctypes.byref(so)) == 0
        assert r_ana(ctypes.byref(sr)) == 0
        d = np.nonzero(bits(so) != bits(sr))[0]
        assert d.size == 0, "analysis bank, frame %d: %d words differ, first at byte %d" % (frame, d.size, 4 * d[0])


def test_bank_parameters_from_frequency_tables(reference):
    """ref_hbe_reinit: what hbe_trans.c:102-222 derives is what tests/hbe_structs.new_state assumes"""
    fn = reference.lib.ref_hbe_reinit
    P16 = ctypes.POINTER(ctypes.c_int16)
    fn.restype, fn.argtypes = ctypes.c_int, [P16, ctypes.c_int, P16, ctypes.c_int, ctypes.POINTER(HbeState)]
    for sb in range(0, 33):
        lo = np.array([sb, min(64, sb + 6), min(64, sb + 14), min(64, sb + 26)], np.int16)
        hi = np.array([sb, min(64, sb + 3), min(64, sb + 6), min(64, sb + 10), min(64, sb + 14), min(64, sb + 26)], np.int16)
        st = HbeState()
        assert fn(lo.ctypes.data_as(P16), 3, hi.ctypes.data_as(P16), 5, ctypes.byref(st)) == 0
        exp = new_state(sb)
        assert (st.synth_size, st.k_start, st.start_band) == (exp.synth_size, exp.k_start, sb)
