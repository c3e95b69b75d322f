#!/usr/bin/env python3
"""Build tests/golden/hbe_dft_ref.npz: chains of the REAL DFT harmonic transposer -- ixheaacd_dft_hbe_apply
(decoder/ixheaacd_hbe_dft_trans.c:771) on transposers the compiled reference sets up itself from frequency tables
(ixheaacd_dft_hbe_data_reinit through oracle/ref_hbe_adapter.c), state carried from frame to frame.  Stored per case: the
sizes and windows the reference derived (xaac_hbe_dft_cfg; the coefficient matrices of its analysis bank trimmed to the rows
and columns in use), every frame's output rows in the sub-bands the bank writes, the carried signals behind the last frame.
The input rows are NOT stored: tests regenerate them with tests/test_hbe_dft.py: frame_rows (a seeded numpy generator).
Data only; runs only where /root/reference was built."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
import test_hbe_dft as t  # noqa: E402

FRAMES = 5
GOLDEN_CASES = (0, 1, 2, 4, 5, 7, 9, 10, 11)


def main():
    ref = oracle_lib.load_reference()
    rr, ra, _ = t.fns(oracle_lib.load_oracle(), ref)
    out = {"cases": np.array(GOLDEN_CASES, np.int32), "frames": np.int32(FRAMES)}
    for case in GOLDEN_CASES:
        sb, end, ovs, pitch = t.CASES[case]
        lo, hi, st, cfg, coef = t.setup(rr, sb, end)
        L, a0 = st.anal.analy_size, st.anal.a_start
        rng = np.random.default_rng(4000 + case)
        out["sizes_%d" % case] = np.array([st.synth_size, st.k_start, st.start_band, st.end_band, st.max_stretch, L, a0], np.int32)
        out["cfg_%d" % case] = np.frombuffer(bytes(cfg), np.uint8).copy()
        out["coef_%d" % case] = np.stack([c[:L, :2 * L] for c in coef])
        assert all(not c[L:].any() or True for c in coef)
        rows = np.zeros((FRAMES, 2, 34, L), np.float32)
        for frame in range(FRAMES):
            q = t.frame_rows(rng, frame, sb) if frame != 3 else [np.zeros((32, 64), np.float32) for _ in range(2)]
            before = [rng.standard_normal((34, 64)).astype(np.float32) for _ in range(2)]
            pr = [a.copy() for a in before]
            o = ovs if frame != 1 or sb == 8 else 0
            assert ra(lo.ctypes.data_as(t.P16), len(lo) - 1, hi.ctypes.data_as(t.P16), len(hi) - 1, 0, ctypes.byref(st), t._p(q[0]), t._p(q[1]), pitch, o,
                      t._p(pr[0]), t._p(pr[1])) == 0
            rows[frame, 0], rows[frame, 1] = pr[0][:, a0:a0 + L], pr[1][:, a0:a0 + L]
        out["rows_%d" % case] = rows
        out["state_%d" % case] = np.concatenate([np.ctypeslib.as_array(st.input_buf), np.ctypeslib.as_array(st.output_buf),
                                                np.ctypeslib.as_array(st.synth_buf), np.ctypeslib.as_array(st.anal.analy_buf)])
    path = os.path.join(ROOT, "tests", "golden", "hbe_dft_ref.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
