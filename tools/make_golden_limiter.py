#!/usr/bin/env python3
"""Build tests/golden/limiter_ref.npz: chains of frames through the REAL ixheaacd_peak_limiter_init /
ixheaacd_peak_limiter_process (oracle/_ref/libref_harness.so) -- inputs, qshift_adj, outputs and the state after
every frame -- for the box that has no reference.  Needs oracle/_ref (only where /root/reference exists)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import limiter_cases as lc  # noqa: E402
import oracle_lib  # noqa: E402

CHAINS = [(1, 8000, 160), (2, 48000, 256), (2, 44100, 1024), (1, 96000, 512)]  # channels, rate, frame_len
FRAMES = 7


def main():
    ref = oracle_lib.load_reference()
    if ref is None:
        raise SystemExit("oracle/_ref not built")
    init, proc, _ = lc.bind(ref.lib, "ref")
    out = {"chains": np.array(CHAINS, np.int32)}
    rng = np.random.default_rng(20260928)
    for ci, (nch, rate, frame_len) in enumerate(CHAINS):
        st = lc.LimiterState()
        init(ctypes.byref(st), nch, rate)
        states = [bytes(st)]
        xin, xout, qs = [], [], []
        for f in range(FRAMES):
            x = lc.signal(rng, lc.KINDS[(f + ci) % len(lc.KINDS)], frame_len, nch)
            q = rng.integers(1, 3, nch).astype(np.int8)
            xin.append(x.copy())
            proc(ctypes.byref(st), x.ctypes.data_as(lc.P32), frame_len, q.ctypes.data_as(lc.P8))
            xout.append(x)
            qs.append(q)
            states.append(bytes(st))
        out["in_%d" % ci] = np.stack(xin)
        out["out_%d" % ci] = np.stack(xout)
        out["q_%d" % ci] = np.stack(qs)
        out["state_%d" % ci] = np.frombuffer(b"".join(states), np.uint8).reshape(FRAMES + 1, -1)
    path = os.path.join(ROOT, "tests", "golden", "limiter_ref.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
