"""Pins the oracle (oracle/oracle_imdct.c) to the real reference
(ixheaacd_imdct_process, decoder/ixheaacd_lpfuncs.c:347) on seeded inputs.
Runs only where oracle/_ref has been built from /root/reference."""
import numpy as np
import pytest

import oracle_lib


@pytest.mark.parametrize("seq", [0, 1, 2, 3])
@pytest.mark.parametrize("pseq", [0, 1, 2, 3])
def test_imdct_all_transitions_bit_exact(oracle, reference, seq, pseq):
    rng = np.random.default_rng(1000 + 4 * seq + pseq)
    spec, ovl = oracle_lib.random_case(rng, 96)
    spec[5] = 0                       # silence -> maximum headroom
    spec[6] = np.int32(-2 ** 31)      # most negative everywhere
    spec[7] = np.int32(2 ** 31 - 1)
    ovl[8] = np.int32(2 ** 31 - 1)
    ovl[9] = np.int32(-2 ** 31)
    for i in range(spec.shape[0]):
        pshape, shape = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        a = reference.imdct_process(spec[i], ovl[i], pseq, pshape, seq, shape)
        b = oracle.imdct_process(spec[i], ovl[i], pseq, pshape, seq, shape)
        assert a[0] == b[0], "qshift_adj"
        assert np.array_equal(a[1], b[1]), "time samples"
        assert np.array_equal(a[2], b[2]), "overlap buffer"
        assert a[3:] == b[3:] == (seq, shape)


def test_imdct_stream_chain_bit_exact(oracle, reference):
    """state carried over a legal window-sequence walk, like a real stream"""
    rng = np.random.default_rng(7)
    nxt = {0: [0, 0, 0, 1], 1: [2, 3], 2: [2, 3], 3: [0, 1]}
    seq, shape = 0, 0
    ro = np.zeros(512, np.int32)
    oo = np.zeros(512, np.int32)
    rs, os_ = (0, 0), (0, 0)
    for frame in range(300):
        spec = rng.integers(-(1 << 17), 1 << 17, 1024).astype(np.int32)
        spec[640:] = 0
        a = reference.imdct_process(spec, ro, rs[0], rs[1], seq, shape)
        b = oracle.imdct_process(spec, oo, os_[0], os_[1], seq, shape)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), frame
        ro, oo, rs, os_ = a[2], b[2], a[3:], b[3:]
        seq = int(rng.choice(nxt[seq]))
        shape = int(rng.integers(0, 2))
