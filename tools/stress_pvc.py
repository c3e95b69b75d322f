#!/usr/bin/env python3
"""Developer tool (GPU box): fresh PVC chains with shifted seeds through xaac_pvc_process_batch against the oracle, CRC per step.
usage: stress_pvc.py [rounds]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pvc_structs as ps  # noqa: E402
import test_pvc as tp  # noqa: E402

if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    fn = ps.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "xo_pvc_process")
    bad = steps = 0
    for r in range(rounds):
        chains = [ps.chain(100000 + 1000 * r + k, 12) for k in range(256)]
        want = np.stack([ps.walk(fn, c) for c in chains])
        got = tp._gpu_walk(chains, 12)
        bad += int((got != want).any(axis=2).sum())
        steps += got.shape[0] * got.shape[1]
    print("pvc stress: %d steps, %d differing" % (steps, bad))
    sys.exit(1 if bad else 0)
