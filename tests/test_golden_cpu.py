"""Oracle vs the committed reference-generated vectors (tests/golden/imdct_ref.npz,
made by tools/make_golden_imdct.py from the compiled reference)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "imdct_ref.npz"))


def test_oracle_reproduces_reference_vectors(oracle):
    g = load_golden()
    assert g["spec"].shape[0] >= 40
    for i in range(g["spec"].shape[0]):
        pseq, pshape, seq, shape = (int(v) for v in g["meta"][i])
        q, y, no, ns, nsh = oracle.imdct_process(g["spec"][i], g["ovl"][i], pseq, pshape, seq, shape)
        assert q == int(g["qadj"][i]), i
        assert np.array_equal(y, g["out"][i]), i
        assert np.array_equal(no, g["ovl_out"][i]), i
        assert (ns, nsh) == (seq, shape)


def test_oracle_batch_matches_single_calls(oracle):
    g = load_golden()
    n = g["spec"].shape[0]
    ics = np.ascontiguousarray(g["meta"][:, 2:4])
    state = np.ascontiguousarray(g["meta"][:, 0:2])
    r = oracle.imdct_batch(g["spec"], ics, g["ovl"], state)
    assert np.array_equal(r["out32"], g["out"])
    assert np.array_equal(r["overlap"], g["ovl_out"])
    assert np.array_equal(r["qshift_adj"], g["qadj"])
    assert np.array_equal(r["state"], ics)
    # PCM16 hand-off, both flavours (api.c:353-366 / peak_limiter.c:324 + api.c:3676)
    for mode in (0, 1):
        pcm = oracle.imdct_batch(g["spec"], ics, g["ovl"], state, pcm_mode=mode)["pcm16"]
        x = g["out"].astype(np.int64)
        sh = g["qadj"].astype(np.int64)[:, None]
        v = x << sh
        if mode == 0:
            v = ((v + 2 ** 31) % 2 ** 32) - 2 ** 31
        else:
            v = np.clip(v, -2 ** 31, 2 ** 31 - 1)
        want = (np.clip(v + 0x8000, -2 ** 31, 2 ** 31 - 1) >> 16).astype(np.int16)
        assert np.array_equal(pcm, want), mode
