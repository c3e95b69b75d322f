#!/usr/bin/env python3
"""Generate tests/golden/imdct_ref.npz: inputs and the REAL reference's outputs
for ixheaacd_imdct_process (decoder/ixheaacd_lpfuncs.c:347), via
oracle/_ref/libref_harness.so.  Runs only in the container that has
/root/reference; the .npz (data only) is committed so that the GPU box, which has
no reference, still checks against reference-produced vectors.

Cases: every (previous, current) window_sequence pair x 2 magnitudes, both
shapes, plus silence / full-scale / sparse spectra; and one 24-frame legal
window-sequence walk with carried state."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402


def main():
    ref = oracle_lib.load_reference()
    assert ref is not None, "build oracle/_ref first (make -f oracle/Makefile.ref)"
    rng = np.random.default_rng(20260928)
    spec, ovl, meta, out, ovl_out, qadj = [], [], [], [], [], []

    def add(s, o, pseq, pshape, seq, shape):
        q, y, no, _, _ = ref.imdct_process(s, o, pseq, pshape, seq, shape)
        spec.append(np.array(s, np.int32)); ovl.append(np.array(o, np.int32))
        meta.append((pseq, pshape, seq, shape)); out.append(y); ovl_out.append(no); qadj.append(q)
        return no

    for pseq in range(4):
        for seq in range(4):
            for mag in (12, 17):
                s = rng.integers(-(1 << mag), 1 << mag, 1024).astype(np.int32)
                s[640:] = 0
                o = rng.integers(-(1 << 14), 1 << 14, 512).astype(np.int32)
                add(s, o, pseq, int(rng.integers(0, 2)), seq, int(rng.integers(0, 2)))
    z = np.zeros(1024, np.int32)
    add(z, np.zeros(512, np.int32), 0, 0, 0, 0)                                   # silence
    add(np.full(1024, 2 ** 31 - 1, np.int32), np.full(512, 2 ** 31 - 1, np.int32), 0, 1, 0, 1)
    add(np.full(1024, -2 ** 31, np.int32), np.full(512, -2 ** 31, np.int32), 2, 1, 2, 0)
    s = z.copy(); s[3] = 1 << 29
    add(s, rng.integers(-(1 << 30), 1 << 30, 512).astype(np.int32), 0, 0, 0, 1)  # q_shift <= 0 branch
    # a legal walk with carried state
    nxt = {0: [0, 0, 1], 1: [2, 3], 2: [2, 3], 3: [0, 1]}
    o = np.zeros(512, np.int32); pseq = pshape = 0; seq = shape = 0
    for _ in range(24):
        s = rng.integers(-(1 << 16), 1 << 16, 1024).astype(np.int32)
        s[512:] //= 64
        o = add(s, o, pseq, pshape, seq, shape)
        pseq, pshape = seq, shape
        seq = int(rng.choice(nxt[seq])); shape = int(rng.integers(0, 2))
    path = os.path.join(ROOT, "tests", "golden", "imdct_ref.npz")
    np.savez_compressed(path, spec=np.stack(spec), ovl=np.stack(ovl), meta=np.array(meta, np.uint8),
                        out=np.stack(out), ovl_out=np.stack(ovl_out), qadj=np.array(qadj, np.int8))
    print(path, os.path.getsize(path), "bytes,", len(spec), "cases")


if __name__ == "__main__":
    main()
