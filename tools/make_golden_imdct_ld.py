#!/usr/bin/env python3
"""Build tests/golden/imdct_ld_ref.npz: chains of the REAL ixheaacd_imdct_process for AAC-LD / AAC-ELD frames (frame_length
512 / 480, object types 23 / 39) run by the compiled reference (ref_imdct_ld_process in oracle/ref_harness.c) with the overlap
and the previous window shape carried.

The spectral lines are NOT stored: tests regenerate them from (config, chain, frame) with chain_spec() below (integer
arithmetic on a counter).  Stored per frame: window shape, CRC32 of the reference's PCM16 output and of its overlap after
the call; the last frame's PCM in full.  Data only; runs only where /root/reference is."""
import ctypes
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden_imdct960 import _mix  # noqa: E402

CONFIGS = [(512, 0), (512, 1), (480, 0), (480, 1)]   # (frame_length, eld)
CHAINS, FRAMES = 12, 30
P32 = ctypes.POINTER(ctypes.c_int32)
P16 = ctypes.POINTER(ctypes.c_int16)


def n_overlap(frame_length, eld):
    return 3 * frame_length if eld else frame_length // 2


def chain_spec(cfg, chain, frame):
    """frame_length spectral lines: noise at a level that walks with (chain, frame); every 5th frame sparse tonal lines, every
    11th silence, every 13th one full-scale line in low-level noise"""
    n = CONFIGS[cfg][0]
    z = _mix((7 << 30) | (cfg << 24) | (chain << 12) | frame, n)
    v = (z >> np.uint64(32)).astype(np.int64) - (1 << 31)
    x = v >> ((3 * chain + 5 * frame) % 29)
    if frame % 11 == 10:
        x[:] = 0
    elif frame % 5 == 4:
        x = np.where((z & np.uint64(63)) == 0, x, 0)
    elif frame % 13 == 12:
        x = v >> 25
        x[int(z[0] % np.uint64(n))] = -(1 << 31) if (int(z[1]) & 1) else (1 << 31) - 1
    return x.astype(np.int32)


def chain_shape(cfg, chain, frame):
    return int(_mix((8 << 30) | (cfg << 24) | (chain << 12) | frame, 1)[0] & np.uint64(1))


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def main():
    import oracle_lib
    ref = oracle_lib.load_reference()
    fn = ref.lib.ref_imdct_ld_process
    fn.restype = ctypes.c_int
    fn.argtypes = [P32, P32, P16, ctypes.c_int, ctypes.c_int, ctypes.c_int, P16, ctypes.c_int]
    shapes = np.zeros((len(CONFIGS), CHAINS, FRAMES), np.uint8)
    crcs = np.zeros((len(CONFIGS), CHAINS, FRAMES, 2), np.uint32)
    last = np.zeros((len(CONFIGS), CHAINS, 512), np.int16)
    for g, (fl, eld) in enumerate(CONFIGS):
        for c in range(CHAINS):
            ov = np.zeros(2048, np.int32)
            ps = np.zeros(1, np.int16)
            for f in range(FRAMES):
                spec = np.zeros(2048, np.int32)
                spec[:fl] = chain_spec(g, c, f)
                pcm = np.zeros(fl, np.int16)
                shapes[g, c, f] = chain_shape(g, c, f)
                q = fn(spec.ctypes.data_as(P32), ov.ctypes.data_as(P32), ps.ctypes.data_as(P16), int(shapes[g, c, f]), fl, 39 if eld else 23,
                       pcm.ctypes.data_as(P16), 1)
                assert q == -2
                crcs[g, c, f] = crc(pcm), crc(ov[:n_overlap(fl, eld)])
            last[g, c, :fl] = pcm
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "imdct_ld_ref.npz"), shape=shapes, crc=crcs, last=last)
    print("wrote", len(CONFIGS), "configs x", CHAINS, "chains x", FRAMES, "frames")


if __name__ == "__main__":
    main()
