#!/usr/bin/env python3
"""Emit tests/golden/hbe_ref.npz: reference-made vectors of the harmonic transposer's polyphase banks.  For every bank
size a chain of frames runs through the compiled reference's ixheaacd_real_synth_filt + ixheaacd_complex_anal_filt
(oracle/ref_hbe_adapter.c), the delay lines carried; per frame the CRC-32 of the whole state after each bank, and the
last frame's time signal and analysis rows in full.  Inputs are regenerated from seeds (chain_input), not stored."""
import ctypes
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hbe_structs import HbeState, new_state  # noqa: E402

START_BANDS = [2, 9, 14, 22, 30]  # synth_size 4, 8, 12, 16, 20
FRAMES = 5
PF = ctypes.POINTER(ctypes.c_float)


def chain_input(chain, frame):
    """QMF columns float32[32, 64] x 2 of frame `frame` of chain `chain`"""
    rng = np.random.default_rng(77000 + 100 * chain + frame)
    a = 2.0 ** rng.integers(0, 16)
    re = (rng.standard_normal((32, 64)) * a).astype(np.float32)
    im = (rng.standard_normal((32, 64)) * a).astype(np.float32)
    if frame == 3:
        re[:], im[:] = 0, 0
    return re, im


def shift_input(st):
    """ixheaacd_qmf_hbe_apply's shift of the time signal before the synthesis bank (hbe_trans.c:235-238)"""
    s = st.synth_size
    buf = np.frombuffer(st, np.float32, 1088, HbeState.input_buf.offset)
    buf[:s] = buf[32 * s:33 * s].copy()


def crc(st):
    return zlib.crc32(bytes(st)) & 0xffffffff


def run(lib, prefix):
    syn, ana = getattr(lib, prefix + "_hbe_real_synth"), getattr(lib, prefix + "_hbe_cplx_anal")
    syn.restype, syn.argtypes = ctypes.c_int, [ctypes.POINTER(HbeState), PF, PF, ctypes.c_int]
    ana.restype, ana.argtypes = ctypes.c_int, [ctypes.POINTER(HbeState)]
    crcs = np.zeros((len(START_BANDS), FRAMES, 2), np.uint32)
    last_time = np.zeros((len(START_BANDS), 1088), np.float32)
    last_rows = np.zeros((len(START_BANDS), 16, 128), np.float32)
    for c, sb in enumerate(START_BANDS):
        st = new_state(sb)
        for f in range(FRAMES):
            re, im = chain_input(c, f)
            shift_input(st)
            assert syn(ctypes.byref(st), re.ctypes.data_as(PF), im.ctypes.data_as(PF), 32) == 0
            crcs[c, f, 0] = crc(st)
            assert ana(ctypes.byref(st)) == 0
            crcs[c, f, 1] = crc(st)
        last_time[c] = np.frombuffer(st, np.float32, 1088, HbeState.input_buf.offset)
        last_rows[c] = np.frombuffer(st, np.float32, 32 * 128, HbeState.qmf_in_buf.offset).reshape(32, 128)[12:28]
    return crcs, last_time, last_rows


P16 = ctypes.POINTER(ctypes.c_int16)
APPLY_FRAMES = 5


def apply_tables():
    """(lo, hi) frequency-band tables of the apply chains: one per bank size with all three stretch factors in use
    where four patches fit, and the two table pairs of the committed HE-AAC stream headers"""
    out = []
    for sb, end, width in ((2, 7, 1), (9, 31, 2), (14, 47, 3), (22, 64, 3), (30, 64, 2)):
        hi = list(range(sb, end, width)) + [end]
        lo = hi[::2] if (len(hi) - 1) % 2 == 0 else [hi[0]] + hi[1::2]
        out.append((np.array(lo, np.int16), np.array(hi, np.int16)))
    out.append((np.array([15, 17, 19, 21, 23, 26, 29, 33, 37, 41], np.int16),
                np.array([15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 26, 27, 29, 31, 33, 35, 37, 39, 41], np.int16)))
    out += [out[1], out[2], out[5]]  # three more chains with a pitch (APPLY_PITCH): the cross-product variants
    return out


APPLY_PITCH = [0, 0, 0, 0, 0, 0, 24, 60, 13]


def apply_input(chain, frame):
    rng = np.random.default_rng(88000 + 100 * chain + frame)
    a = 2.0 ** rng.integers(2, 14)
    re = (rng.standard_normal((32, 64)) * a).astype(np.float32)
    im = (rng.standard_normal((32, 64)) * a).astype(np.float32)
    if frame == 2:
        re[:], im[:] = 0, 0
    if chain >= 6 and frame != 2:  # the pitch chains: core-band tones, so that cross products are taken
        re *= np.float32(1e-3)
        im *= np.float32(1e-3)
        for k in range(1, 9, 2):
            ph = rng.uniform(0, 6.28)
            re[:, k] += (a * np.cos(ph + 0.9 * k * np.arange(32))).astype(np.float32)
            im[:, k] += (a * np.sin(ph + 0.9 * k * np.arange(32))).astype(np.float32)
    return re, im


def apply_state(ref_lib, lo, hi):
    fn = ref_lib.ref_hbe_reinit
    fn.restype, fn.argtypes = ctypes.c_int, [P16, ctypes.c_int, P16, ctypes.c_int, ctypes.POINTER(HbeState)]
    st = HbeState()
    assert fn(lo.ctypes.data_as(P16), len(lo) - 1, hi.ctypes.data_as(P16), len(hi) - 1, ctypes.byref(st)) == 0
    return st


def apply_params(st):
    return [st.synth_size, st.k_start, st.start_band, st.end_band] + list(st.x_over_qmf) + [st.max_stretch]


def state_from_params(par):
    st = HbeState()
    st.synth_size, st.k_start, st.start_band, st.end_band = [int(v) for v in par[:4]]
    for q in range(6):
        st.x_over_qmf[q] = int(par[4 + q])
    st.max_stretch = int(par[10])
    return st


def run_apply_reference(ref_lib):
    fn = ref_lib.ref_hbe_apply
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(HbeState), P16, ctypes.c_int, P16, ctypes.c_int, PF, PF, ctypes.c_int, PF, PF]
    tabs = apply_tables()
    crcs = np.zeros((len(tabs), APPLY_FRAMES, 3), np.uint32)
    params = np.zeros((len(tabs), 11), np.int32)
    last_pv = np.zeros((len(tabs), 2, 32, 64), np.float32)
    for c, (lo, hi) in enumerate(tabs):
        st = apply_state(ref_lib, lo, hi)
        params[c] = apply_params(st)
        for f in range(APPLY_FRAMES):
            re, im = apply_input(c, f)
            pv = np.full((2, 32, 64), 7.5, np.float32)
            assert fn(ctypes.byref(st), lo.ctypes.data_as(P16), len(lo) - 1, hi.ctypes.data_as(P16), len(hi) - 1,
                      re.ctypes.data_as(PF), im.ctypes.data_as(PF), APPLY_PITCH[c], pv[0].ctypes.data_as(PF), pv[1].ctypes.data_as(PF)) == 0
            crcs[c, f] = [crc(st), zlib.crc32(pv[0].tobytes()) & 0xffffffff, zlib.crc32(pv[1].tobytes()) & 0xffffffff]
        last_pv[c] = pv
    return crcs, params, last_pv


DFT_TABLES = [(9, 31), (10, 35), (22, 64), (6, 61)]   # analy_size 24, 28, 44, 60
DFT_FRAMES = 3


def dft_tables(kind):
    sb, end = DFT_TABLES[kind]
    hi = list(range(sb, end, 3)) + [end]
    lo = hi[::2] if (len(hi) - 1) % 2 == 0 else [hi[0]] + hi[1::2]
    return np.array(lo, np.int16), np.array(hi, np.int16)


def dft_input(chain, frame):
    """(time signal float32[4096], rows before the call float32[2][34][64])"""
    rng = np.random.default_rng(99000 + 100 * chain + frame)
    t = (rng.standard_normal(4096) * 2.0 ** rng.integers(0, 12)).astype(np.float32)
    return t, rng.standard_normal((2, 34, 64)).astype(np.float32)


def run_dft_reference(ref_lib):
    """the DFT transposer's analysis bank on the reference's own coefficient matrices: per chain the sizes it derives, its
    matrices (as float16-exact? no: float32, stored), and per frame the CRCs of delay line + both row blocks"""
    from hbe_structs import HbeDftState
    fn = ref_lib.ref_hbe_dft_anal
    fn.restype = ctypes.c_int
    fn.argtypes = [P16, ctypes.c_int, P16, ctypes.c_int, ctypes.POINTER(HbeDftState), PF, ctypes.c_int, PF, PF, PF, PF]
    n = len(DFT_TABLES)
    sizes = np.zeros((n, 2), np.int32)
    coef = np.zeros((n, 2, 64, 128), np.float32)
    crcs = np.zeros((n, DFT_FRAMES, 3), np.uint32)
    for c in range(n):
        lo, hi = dft_tables(c)
        st = HbeDftState()
        for f in range(DFT_FRAMES):
            t, q = dft_input(c, f)
            assert fn(lo.ctypes.data_as(P16), len(lo) - 1, hi.ctypes.data_as(P16), len(hi) - 1, ctypes.byref(st), t.ctypes.data_as(PF), 4096,
                      coef[c, 0].ctypes.data_as(PF), coef[c, 1].ctypes.data_as(PF), q[0].ctypes.data_as(PF), q[1].ctypes.data_as(PF)) == 0
            crcs[c, f] = [zlib.crc32(bytes(st)) & 0xffffffff, zlib.crc32(q[0].tobytes()) & 0xffffffff, zlib.crc32(q[1].tobytes()) & 0xffffffff]
        sizes[c] = [st.analy_size, st.a_start]
    return sizes, coef, crcs


if __name__ == "__main__":
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_harness.so"))
    crcs, last_time, last_rows = run(ref, "ref")
    acrc, apar, apv = run_apply_reference(ref)
    dsz, dcoef, dcrc = run_dft_reference(ref)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hbe_ref.npz"), crc=crcs, last_time=last_time, last_rows=last_rows,
                        apply_crc=acrc, apply_params=apar, apply_last_pv=apv, dft_sizes=dsz, dft_coef=dcoef, dft_crc=dcrc)
    print("dft chains", dsz.tolist())
    print("apply chains", len(apar), "params", apar.tolist())
    print("chains", len(START_BANDS), "frames", FRAMES, "crc[0]", crcs[0].tolist())
