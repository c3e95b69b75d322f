"""Pins the peak-limiter oracle (oracle/oracle_limiter.cpp over libxaac_amd/csrc/limiter.h) to the reference's
own ixheaacd_peak_limiter_init / _process (oracle/_ref/libref_harness.so): outputs and the whole state, frame
after frame.  CPU only."""
import ctypes

import numpy as np
import pytest

import limiter_cases as lc


@pytest.mark.parametrize("nch,rate", [(1, 48000), (2, 44100), (2, 48000), (2, 96000), (1, 8000), (2, 32000),
                                       (6, 48000), (2, 22050)])
def test_init_and_chains(oracle, reference, nch, rate):
    o_init, o_proc, _ = lc.bind(oracle.lib, "xo")
    r_init, r_proc, _ = lc.bind(reference.lib, "ref")
    so, sr = lc.LimiterState(), lc.LimiterState()
    assert o_init(ctypes.byref(so), nch, rate) == r_init(ctypes.byref(sr), nch, rate) == int(5.0 * rate / 1000)
    assert lc.state_view(so) == lc.state_view(sr)
    rng = np.random.default_rng(1000 * nch + rate)
    for frame in range(24):
        kind = lc.KINDS[(frame * 5 + nch) % len(lc.KINDS)]
        frame_len = 1024 if frame % 7 != 6 else int(rng.integers(1, 1025))
        x = lc.signal(rng, kind, frame_len, nch)
        q = rng.integers(0, 3, nch).astype(np.int8) if frame % 5 == 4 else np.full(nch, 1 + frame % 2, np.int8)
        xo, xr = x.copy(), x.copy()
        o_proc(ctypes.byref(so), xo.ctypes.data_as(lc.P32), frame_len, q.ctypes.data_as(lc.P8))
        r_proc(ctypes.byref(sr), xr.ctypes.data_as(lc.P32), frame_len, q.ctypes.data_as(lc.P8))
        assert np.array_equal(xo, xr), (frame, kind)
        assert lc.state_view(so) == lc.state_view(sr), (frame, kind)
    assert so.min_gain <= 1.0


def test_limiter_off_branch(oracle, reference):
    """limiter_on = 0 with a fully released gain: the plain-delay branch (peak_limiter.c:288-300)"""
    o_init, o_proc, _ = lc.bind(oracle.lib, "xo")
    r_init, r_proc, _ = lc.bind(reference.lib, "ref")
    so, sr = lc.LimiterState(), lc.LimiterState()
    o_init(ctypes.byref(so), 2, 48000)
    r_init(ctypes.byref(sr), 2, 48000)
    rng = np.random.default_rng(5)
    for st in (so, sr):
        st.limiter_on = 0
        st.pre_smoothed_gain = 0.0
    q = np.array([1, 2], np.int8)
    for frame in range(3):
        x = lc.signal(rng, "loud" if frame else "quiet", 1024, 2)
        xo, xr = x.copy(), x.copy()
        o_proc(ctypes.byref(so), xo.ctypes.data_as(lc.P32), 1024, q.ctypes.data_as(lc.P8))
        r_proc(ctypes.byref(sr), xr.ctypes.data_as(lc.P32), 1024, q.ctypes.data_as(lc.P8))
        assert np.array_equal(xo, xr)
        assert lc.state_view(so) == lc.state_view(sr)


def test_batch_entries_agree(oracle, reference):
    _, _, o_batch = lc.bind(oracle.lib, "xo")
    o_init, _, _ = lc.bind(oracle.lib, "xo")
    _, _, r_batch = lc.bind(reference.lib, "ref")
    n, nch = 6, 2
    rng = np.random.default_rng(9)
    so = (lc.LimiterState * n)()
    for i in range(n):
        o_init(ctypes.byref(so[i]), nch, 48000)
    sr = (lc.LimiterState * n)()
    ctypes.memmove(sr, so, ctypes.sizeof(so))
    for frame in range(3):
        x = np.concatenate([lc.signal(rng, lc.KINDS[(i + frame) % len(lc.KINDS)], 1024, nch) for i in range(n)])
        q = rng.integers(1, 3, n * nch).astype(np.int8)
        xo, xr = x.copy(), x.copy()
        po, pr = np.zeros(n * 1024 * nch, np.int16), np.zeros(n * 1024 * nch, np.int16)
        o_batch(n, 1024, nch, xo.ctypes.data_as(lc.P32), 1024 * nch, q.ctypes.data_as(lc.P8), so, po.ctypes.data_as(lc.P16))
        r_batch(n, 1024, nch, xr.ctypes.data_as(lc.P32), 1024 * nch, q.ctypes.data_as(lc.P8), sr, pr.ctypes.data_as(lc.P16))
        assert np.array_equal(xo, xr) and np.array_equal(po, pr)
        for i in range(n):
            assert lc.state_view(so[i]) == lc.state_view(sr[i])
