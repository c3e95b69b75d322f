"""The peak-limiter oracle against tests/golden/limiter_ref.npz (outputs + states of the REAL reference,
tools/make_golden_limiter.py).  CPU only, no reference needed."""
import ctypes
import os

import numpy as np

import limiter_cases as lc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "limiter_ref.npz")


def state_from(row):
    st = lc.LimiterState()
    ctypes.memmove(ctypes.byref(st), row.ctypes.data, ctypes.sizeof(st))
    return st


def test_oracle_matches_reference_vectors(oracle):
    init, proc, _ = lc.bind(oracle.lib, "xo")
    g = np.load(GOLD)
    for ci, (nch, rate, frame_len) in enumerate(g["chains"]):
        st = lc.LimiterState()
        assert init(ctypes.byref(st), int(nch), int(rate)) == int(5.0 * rate / 1000)
        states = np.ascontiguousarray(g["state_%d" % ci])
        assert lc.state_view(st) == lc.state_view(state_from(states[0]))
        for f in range(g["in_%d" % ci].shape[0]):
            x = np.ascontiguousarray(g["in_%d" % ci][f])
            q = np.ascontiguousarray(g["q_%d" % ci][f])
            proc(ctypes.byref(st), x.ctypes.data_as(lc.P32), int(frame_len), q.ctypes.data_as(lc.P8))
            assert np.array_equal(x, g["out_%d" % ci][f]), (ci, f)
            assert lc.state_view(st) == lc.state_view(state_from(states[f + 1])), (ci, f)


def test_product_init_matches_reference_vectors():
    """xaac_peak_limiter_init runs on the host: checked here without a GPU"""
    import libxaac_amd
    g = np.load(GOLD)
    for ci, (nch, rate, _) in enumerate(g["chains"]):
        st, delay = libxaac_amd.peak_limiter_init(int(nch), int(rate))
        assert delay == int(5.0 * rate / 1000)
        assert lc.state_view(st) == lc.state_view(state_from(np.ascontiguousarray(g["state_%d" % ci])[0]))
    for bad in ((0, 48000), (9, 48000), (2, 100), (2, 192000)):
        try:
            libxaac_amd.peak_limiter_init(*bad)
        except libxaac_amd.XaacError:
            continue
        raise AssertionError("accepted %r" % (bad,))
