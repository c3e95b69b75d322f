#!/usr/bin/env python3
"""Build tests/golden/esbr_qmf_ref.npz: chains of the REAL eSBR (Path A) QMF banks -- ixheaacd_esbr_analysis_filt_block
(sbr_dec.c:185) and the bank loop of ixheaacd_esbr_synthesis_filt_block (sbr_dec.c:447) -- run by the compiled reference
(oracle/_ref/libref_harness.so through oracle/ref_sbr_adapter.c) with its state carried from frame to frame.

The float inputs of a frame are NOT stored: tests regenerate them from (kind, chain, frame) with chain_input() below, a
counter-based generator written out in integer arithmetic (no library RNG) whose last step, an int -> float conversion
and a multiplication by a power of two, is exact.  Stored per frame: CRC32 of the reference's float output words and of
its ring after the call, the positions; the last frame's outputs in full.  Data only; runs only where /root/reference is."""
import ctypes
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

CHAINS, FRAMES = 12, 9
# level exponents: value = 24-bit uniform integer * 2^e.  analysis inputs are PCM-scaled floats, synthesis inputs QMF samples
ANA_EXP = (-8, -9, -12, -16, -23, -8)      # up to +-32768 (full scale), down to +-1
SYN_EXP = (-6, -10, -14, -20, -3, -6)


def _mix(base, n):
    z = (np.uint64(base) * np.uint64(4096) + np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.int64) - (1 << 23)      # 24-bit signed: exact in a float


def chain_input(kind, chain, frame):
    """kind 0: 1024 core samples; kind 1: (re, im) rows [32][64] with the bands above a chain-dependent limit empty"""
    e = (ANA_EXP, SYN_EXP)[kind][(chain + frame) % 6]
    base = (kind << 24) | (chain << 12) | frame
    if kind == 0:
        v = _mix(base, 1024).astype(np.float32) * np.float32(2.0 ** e)
        if frame % 4 == 3:                                         # plateaus at full level and sign flips
            v[::7] = np.float32((1 << 23) * 2.0 ** e)
            v[1::7] = np.float32(-(1 << 23) * 2.0 ** e)
        return np.ascontiguousarray(v)
    v = _mix(base, 4096).astype(np.float32) * np.float32(2.0 ** e)
    re, im = v[:2048].reshape(32, 64).copy(), v[2048:].reshape(32, 64).copy()
    top = (64, 40, 48, 32)[chain % 4]
    re[:, top:] = 0
    im[:, top:] = 0
    return re, im


NB_BANKS = ((24, 32), (16, 64))   # (analysis channels, time slots): 8:3 and 4:1 SBR


def chain_input_nb(nb, chain, frame):
    """core samples of a frame of the 24- / 16-channel bank: the first nb * slots values of the 32-channel chain's frame"""
    return np.ascontiguousarray(chain_input(0, chain + (nb << 4), frame)[:nb * dict(NB_BANKS)[nb]])


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def main():
    import oracle_lib
    import test_esbr_qmf_oracle_vs_reference as t
    ref = oracle_lib.load_reference()
    ra, rs = t._bind(ref.lib, "ref")
    out = {}
    a_crc = np.zeros((CHAINS, FRAMES, 4), np.uint32)    # re, im, ring, (pos << 16 | win)
    s_crc = np.zeros((CHAINS, FRAMES, 3), np.uint32)    # out, ring, (drc << 16 | filt)
    a_last = np.zeros((CHAINS, 2, 32, 64), np.float32)
    s_last = np.zeros((CHAINS, 2048), np.float32)
    for c in range(CHAINS):
        ring, pos, win = np.zeros(320, np.int32), 0, 0
        for f in range(FRAMES):
            re, im, pos, win = t.ana(ra, chain_input(0, c, f), ring, pos, win)
            a_crc[c, f] = crc(re), crc(im), crc(ring), (pos << 16) | win
        a_last[c, 0], a_last[c, 1] = re, im
        ring, drc, filt = np.zeros(1280, np.int32), 0, 0
        for f in range(FRAMES):
            re, im = chain_input(1, c, f)
            o, drc, filt = t.syn(rs, re, im, ring, drc, filt)
            s_crc[c, f] = crc(o), crc(ring), (drc << 16) | filt
        s_last[c] = o
    for nb, slots in NB_BANKS:   # sbr_dec.c:213-236: the banks of 8:3 and 4:1 SBR, the same function on their own tables
        fn = t._bind_nb(ref.lib, "ref")
        n_crc = np.zeros((CHAINS, FRAMES, 4), np.uint32)
        n_last = np.zeros((CHAINS, 2, slots, 64), np.float32)
        for c in range(CHAINS):
            ring, pos, win = np.zeros(320, np.int32), 0, 0
            for f in range(FRAMES):
                re, im, pos, win = t.ana_nb(fn, chain_input_nb(nb, c, f), nb, slots, ring, pos, win)
                n_crc[c, f] = crc(re), crc(im), crc(ring), (pos << 16) | win
            n_last[c, 0], n_last[c, 1] = re, im
        out["ana%d_crc" % nb], out["ana%d_last" % nb] = n_crc, n_last
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "esbr_qmf_ref.npz"), ana_crc=a_crc, syn_crc=s_crc,
                        ana_last=a_last, syn_last=s_last, **out)
    print("wrote", CHAINS, "chains x", FRAMES, "frames per bank")


if __name__ == "__main__":
    main()
