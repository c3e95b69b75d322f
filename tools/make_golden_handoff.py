#!/usr/bin/env python3
"""Generate tests/golden/handoff_ref.npz: the WORD32 -> WORD16 hand-off behind the IMDCT by the
REAL reference's own code, via oracle/_ref/libref_harness.so: ref_pcm_handoff calls ixheaacd_allocate_sbr_scr
(api.c:337-370, mode 1 = core -> SBR) and ixheaacd_scale_adjust + ixheaac_round16 (peak_limiter.c:324 + api.c:3676-3681,
mode 0 = AAC-LC with the limiter off).  For every case of tests/golden/imdct_ref.npz (the reference's own IMDCT outputs
and qshift_adj) as a mono block and, pairwise, as interleaved stereo blocks; plus saturation corners.  Runs only where
/root/reference exists; the .npz (data only) is committed."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402


def handoff(lib, x, q, mode):
    """x int32[1024, nch] interleaved, q int8[nch] -> int16[1024, nch]"""
    nch = x.shape[1]
    buf = np.ascontiguousarray(x, np.int32).copy()
    qq = np.ascontiguousarray(q, np.int8).copy()
    out = np.zeros((1024, nch), np.int16)
    lib.ref_pcm_handoff(buf.ctypes.data_as(ctypes.c_void_p), qq.ctypes.data_as(ctypes.c_void_p), nch, mode,
                        out.ctypes.data_as(ctypes.c_void_p))
    return out


def main():
    ref = oracle_lib.load_reference()
    assert ref is not None, "build oracle/_ref first (make -f oracle/Makefile.ref)"
    lib = ref.lib
    lib.ref_pcm_handoff.restype = None
    lib.ref_pcm_handoff.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    g = np.load(os.path.join(ROOT, "tests", "golden", "imdct_ref.npz"))
    rng = np.random.default_rng(8)
    xs, qs = list(g["out"]), list(g["qadj"])
    for amp in (30, 31):       # corners: values that saturate in shl32_sat / wrap in the LC multiply
        x = rng.integers(-(1 << amp), (1 << amp) - 1, 1024).astype(np.int64)
        x[:8] = [2 ** 31 - 1, -2 ** 31, 2 ** 30, -2 ** 30 - 1, 0x7fff7fff, -0x7fff8001, 0x3fffc000, 0x3fffbfff]
        xs.append(x.astype(np.int32)); qs.append(int(rng.integers(1, 3)))
    xs, qs = np.stack(xs), np.array(qs, np.int8)
    n = xs.shape[0] & ~1
    mono = {m: np.stack([handoff(lib, xs[i][:, None], qs[i:i + 1], m)[:, 0] for i in range(xs.shape[0])]) for m in (0, 1)}
    stereo = {m: np.stack([handoff(lib, np.stack([xs[i], xs[i + 1]], 1), qs[i:i + 2], m) for i in range(0, n, 2)])
              for m in (0, 1)}
    path = os.path.join(ROOT, "tests", "golden", "handoff_ref.npz")
    np.savez_compressed(path, x=xs, q=qs, mono_lc=mono[0], mono_sbr=mono[1], stereo_lc=stereo[0], stereo_sbr=stereo[1])
    print(path, os.path.getsize(path), "bytes,", xs.shape[0], "blocks")


if __name__ == "__main__":
    main()
