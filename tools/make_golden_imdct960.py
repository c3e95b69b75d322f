#!/usr/bin/env python3
"""Build tests/golden/imdct960_ref.npz: chains of the REAL ixheaacd_imdct_process with frame_length 960
(decoder/ixheaacd_lpfuncs.c:347) run by the compiled reference (ref_imdct960_process in oracle/ref_harness.c) with the
overlap and the previous window sequence / shape carried along legal window-sequence walks.

The 960 spectral lines of a frame are NOT stored: tests regenerate them from (chain, frame) with chain_spec() below
(integer arithmetic on a counter, no library RNG).  Stored per frame: window sequence, shape, qshift_adj, CRC32 of the
reference's WORD32 output and of its overlap after the call; the last frame's output and overlap in full.  Data only;
runs only where /root/reference is."""
import ctypes
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

CHAINS, FRAMES = 32, 40
NEXT = {0: (0, 1), 1: (2, 3), 2: (2, 3), 3: (0, 1)}
P32 = ctypes.POINTER(ctypes.c_int32)
P16 = ctypes.POINTER(ctypes.c_int16)


def _mix(base, n):
    z = (np.uint64(base) * np.uint64(4096) + np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def chain_spec(chain, frame):
    """960 spectral lines: uniform noise at a level that walks with (chain, frame); every 5th frame sparse tonal lines,
    every 11th silence, every 13th one full-scale line in low-level noise"""
    z = _mix((5 << 30) | (chain << 12) | frame, 960)
    v = (z >> np.uint64(32)).astype(np.int64) - (1 << 31)
    x = v >> ((3 * chain + 5 * frame) % 29)
    if frame % 11 == 10:
        x[:] = 0
    elif frame % 5 == 4:
        x = np.where((z & np.uint64(63)) == 0, x, 0)
    elif frame % 13 == 12:
        x = v >> 25
        x[int(z[0] % np.uint64(960))] = -(1 << 31) if (int(z[1]) & 1) else (1 << 31) - 1
    return x.astype(np.int32)


def chain_side(chain, frame, prev_seq):
    """(window sequence, shape) of the frame: a legal successor of the previous frame's sequence"""
    z = _mix((6 << 30) | (chain << 12) | frame, 2)
    nxt = NEXT[prev_seq]
    return int(nxt[int(z[1] >> np.uint64(8)) % len(nxt)]), int(z[0] & np.uint64(1))


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def main():
    import oracle_lib
    ref = oracle_lib.load_reference()
    fn = ref.lib.ref_imdct960_process
    fn.restype = ctypes.c_int
    fn.argtypes = [P32, P32, P16, P16, ctypes.c_int, ctypes.c_int, P32, ctypes.c_int]
    side = np.zeros((CHAINS, FRAMES, 3), np.int8)
    crcs = np.zeros((CHAINS, FRAMES, 2), np.uint32)
    last = np.zeros((CHAINS, 2, 960), np.int32)
    for c in range(CHAINS):
        ov = np.zeros(512, np.int32)
        ps, pw = np.zeros(1, np.int16), np.zeros(1, np.int16)
        for f in range(FRAMES):
            seq, shape = chain_side(c, f, int(ps[0]))
            spec = np.zeros(1024, np.int32)
            spec[:960] = chain_spec(c, f)
            out = np.zeros(960, np.int32)
            q = fn(spec.ctypes.data_as(P32), ov.ctypes.data_as(P32), ps.ctypes.data_as(P16), pw.ctypes.data_as(P16), seq, shape,
                   out.ctypes.data_as(P32), 1)
            side[c, f] = seq, shape, q
            crcs[c, f] = crc(out), crc(ov[:480])
        last[c, 0], last[c, 1, :480] = out, ov[:480]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "imdct960_ref.npz"), side=side, crc=crcs, last=last)
    print("wrote", CHAINS, "chains x", FRAMES, "frames; sequences seen:", np.bincount(side[:, :, 0].ravel(), minlength=4))


if __name__ == "__main__":
    main()
