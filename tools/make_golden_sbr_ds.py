#!/usr/bin/env python3
"""Build tests/golden/sbr_ds_ref.npz: what the REAL ixheaacd_sbr_dec produces with the down-sampled synthesis bank
(32 channels) on the chains of tests/sbr_ds_cases.py -- outputs, return codes and the synthesis ring after the last
frame -- for the box that has no reference.  Needs oracle/_ref."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
import sbr_ds_cases as ds  # noqa: E402
from test_sbr_dsample import reference_call  # noqa: E402

ref = oracle_lib.load_reference()
if ref is None:
    raise SystemExit("oracle/_ref not built")
out = {}
for low_pow, key in ((1, "lp"), (0, "hq")):
    o, rc, st = ds.run_chain(reference_call(ref, low_pow), low_pow, ds.cases(low_pow))
    out[key + "_out"] = o
    out[key + "_rc"] = np.array(rc, np.int32)
    out[key + "_ring"] = np.stack([np.ctypeslib.as_array(s.syn_ring)[:640].copy() for s in st])
    out[key + "_pos"] = np.array([(s.syn_drc_offset, s.syn_phase) for s in st], np.int32)
path = os.path.join(ROOT, "tests", "golden", "sbr_ds_ref.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes")
