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


if __name__ == "__main__":
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_harness.so"))
    crcs, last_time, last_rows = run(ref, "ref")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hbe_ref.npz"), crc=crcs, last_time=last_time, last_rows=last_rows)
    print("chains", len(START_BANDS), "frames", FRAMES, "crc[0]", crcs[0].tolist())
