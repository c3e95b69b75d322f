#!/usr/bin/env python3
"""tests/golden/pvc_ref.npz: what the REAL reference (oracle/_ref/libref_harness.so: ref_pvc_process = ixheaacd_qmf_enrg_calc +
ixheaacd_pvc_process as ixheaacd_sbr_dec calls them) makes of the seeded frame chains of tests/pvc_structs.py: per frame the
return code and the CRC32 of the 1024 output floats and of the carried state.  The inputs are regenerated from the seeds by
the tests; nothing of the reference is stored but these numbers."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pvc_structs as ps  # noqa: E402

SEEDS, FRAMES = list(range(5000, 5048)), 24

if __name__ == "__main__":
    ref = ps.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_harness.so")), "ref_pvc_process")
    res = np.stack([ps.walk(ref, ps.chain(s, FRAMES)) for s in SEEDS])
    assert not res[:, :, 0].any()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pvc_ref.npz"), seeds=np.array(SEEDS), frames=np.array(FRAMES), res=res)
    print("pvc_ref.npz:", res.shape, "steps:", res.shape[0] * res.shape[1])
