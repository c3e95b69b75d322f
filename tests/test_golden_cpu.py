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


def test_pcm_handoff_matches_the_references_own_code(oracle):
    """row a8: both PCM16 hand-off flavours of the oracle against tests/golden/handoff_ref.npz, which
    tools/make_golden_handoff.py made by calling ixheaacd_allocate_sbr_scr (api.c:337-370) and ixheaacd_scale_adjust +
    ixheaac_round16 (peak_limiter.c:324, api.c:3676-3681) themselves"""
    import ctypes
    h = np.load(os.path.join(ROOT, "tests", "golden", "handoff_ref.npz"))
    P32, P16 = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int16)
    oracle.lib.xo_pcm16.restype = None
    oracle.lib.xo_pcm16.argtypes = [P32, ctypes.c_int, P16, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    for mode, key in ((0, "mono_lc"), (1, "mono_sbr")):
        for i in range(h["x"].shape[0]):
            x = np.ascontiguousarray(h["x"][i])
            out = np.zeros(1024, np.int16)
            oracle.lib.xo_pcm16(x.ctypes.data_as(P32), 1, out.ctypes.data_as(P16), 1, 1024, int(h["q"][i]), mode)
            assert np.array_equal(out, h[key][i]), (mode, i)
    oracle.lib.xo_pcm16_block.restype = None
    for mode, key in ((0, "stereo_lc"), (1, "stereo_sbr")):       # interleaved, in place like the reference
        for p in range(h[key].shape[0]):
            blk = np.ascontiguousarray(np.stack([h["x"][2 * p], h["x"][2 * p + 1]], 1))
            q = np.ascontiguousarray(h["q"][2 * p:2 * p + 2])
            out = np.zeros((1024, 2), np.int16)
            oracle.lib.xo_pcm16_block(blk.ctypes.data_as(ctypes.c_void_p), q.ctypes.data_as(ctypes.c_void_p), 2, mode,
                                      out.ctypes.data_as(ctypes.c_void_p))
            assert np.array_equal(out, h[key][p]), (mode, p)
    # and the in-place order matters: the stereo SBR hand-off is NOT the per-channel conversion (ch 1, samples < 512)
    diff = sum(int(np.any(h["stereo_sbr"][p][:, 1] != h["mono_sbr"][2 * p + 1])) for p in range(h["stereo_sbr"].shape[0]))
    same = all(np.array_equal(h["stereo_sbr"][p][:, 0], h["mono_sbr"][2 * p]) and
               np.array_equal(h["stereo_sbr"][p][512:, 1], h["mono_sbr"][2 * p + 1][512:]) for p in range(h["stereo_sbr"].shape[0]))
    assert same and diff > 0
