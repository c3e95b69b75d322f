"""Pins the SBR QMF oracle (oracle/oracle_qmf.cpp + libxaac_amd/csrc/sbr_qmf.h) to the compiled
reference, symbol by symbol and bank by bank, persistent ring state included.  Needs oracle/_ref."""
import ctypes

import numpy as np
import pytest

import oracle_lib
from oracle_lib import P16, P32, _p


def rnd(rng, n, mag):
    return rng.integers(-(1 << mag), 1 << mag, n).astype(np.int32)


def test_transform_pieces_bit_exact(oracle, reference):
    ref, orc = reference.lib, oracle.lib
    rng = np.random.default_rng(31)
    for trial in range(400):
        mag = int(rng.integers(2, 31))
        x = rnd(rng, 32, mag); ya = np.zeros(32, np.int32); yb = np.zeros(32, np.int32)
        ref.ref_postradix4(_p(ya, P32), _p(x.copy(), P32)); orc.xo_postradix4(_p(yb, P32), _p(x.copy(), P32))
        assert np.array_equal(ya, yb)
        x = rnd(rng, 64, mag); ya = np.zeros(64, np.int32); yb = np.zeros(64, np.int32)
        ref.ref_postradix2(_p(ya, P32), _p(x.copy(), P32)); orc.xo_postradix2(_p(yb, P32), _p(x.copy(), P32))
        assert np.array_equal(ya, yb)
        x = rnd(rng, 64, mag); oa = np.zeros(32, np.int32); ob = np.zeros(32, np.int32)
        ref.ref_dct3_32(_p(x.copy(), P32), _p(oa, P32)); orc.xo_dct3_32(_p(x.copy(), P32), _p(ob, P32))
        assert np.array_equal(oa, ob), "dct3_32"
        for m in (16, 32):
            s = rnd(rng, 128, mag); sa = s.copy(); sb = s.copy()
            ref.ref_cos_sin_mod(_p(sa, P32), m); orc.xo_cos_sin_mod(_p(sb, P32), m)
            idx = np.r_[0:2 * m, 64:64 + 2 * m]
            assert np.array_equal(sa[idx], sb[idx]), "cos_sin_mod %d" % m
        x = rnd(rng, 64, mag); ba = np.zeros(160, np.int16); bb = np.zeros(160, np.int16)
        ref.ref_inv_modulation_lp(_p(x.copy(), P32), _p(ba, P16)); orc.xo_dct2_64_lp(_p(x.copy(), P32), _p(bb, P16))
        assert np.array_equal(ba[:128], bb[:128]), "LP synthesis slot"
        s = rnd(rng, 128, mag); sh = int(rng.integers(0, 12)); ba = np.zeros(128, np.int16); bb = np.zeros(128, np.int16)
        ref.ref_synth_hq_slot(_p(s.copy(), P32), _p(ba, P16), sh); orc.xo_synth_hq_slot(_p(s.copy(), P32), _p(bb, P16), sh)
        assert np.array_equal(ba, bb), "HQ synthesis slot"


@pytest.mark.parametrize("low_pow", [1, 0])
@pytest.mark.parametrize("stride", [1, 2])
def test_analysis_bank_with_state(oracle, reference, low_pow, stride):
    ref, orc = reference.lib, oracle.lib
    rng = np.random.default_rng(7 + low_pow)
    st = oracle_lib.QmfAnaState()
    ring = np.zeros(320, np.int16); wr = np.zeros(1, np.int16); ph = np.zeros(1, np.int16)
    ss = 64 if low_pow else 128
    for f in range(16):
        amp = 32768 if f % 3 else 200
        pcm = rng.integers(-amp, amp, 1024 * stride).astype(np.int16)
        if f == 5:
            pcm[:] = -32768
        qa = np.zeros(32 * ss, np.int32); qb = np.zeros(32 * ss, np.int32)
        ref.ref_qmf_analysis(_p(pcm, P16), stride, _p(ring, P16), _p(wr, P16), _p(ph, P16), low_pow, 32, _p(qa, P32), ss)
        orc.xo_qmf_analysis(_p(pcm, P16), stride, ctypes.byref(st), low_pow, 32, _p(qb, P32), ss)
        assert np.array_equal(qa, qb), f
        assert np.array_equal(np.array(st.ring[:]), ring) and st.wr == wr[0] and st.phase == ph[0], f


@pytest.mark.parametrize("low_pow", [1, 0])
@pytest.mark.parametrize("stride", [1, 2])
def test_synthesis_bank_with_state(oracle, reference, low_pow, stride):
    ref, orc = reference.lib, oracle.lib
    rng = np.random.default_rng(17 + low_pow)
    st = oracle_lib.QmfSynState()
    ring = np.zeros(1280, np.int16); d = np.zeros(1, np.int16); ph = np.zeros(1, np.int16)
    ss = 64 if low_pow else 128
    for f in range(16):
        mag = int(rng.integers(8, 30))
        q = rnd(rng, 32 * ss, mag)
        sf = np.array([rng.integers(-12, 4), rng.integers(-12, 4), rng.integers(-12, 4), rng.integers(-8, 0)], np.int16)
        lsb = int(rng.integers(8, 33)); usb = int(rng.integers(lsb, 65))
        pa = np.zeros(2048 * stride, np.int16); pb = np.zeros(2048 * stride, np.int16)
        ref.ref_qmf_synthesis(_p(q.copy(), P32), ss, _p(sf, P16), lsb, usb, 6, _p(ring, P16), _p(d, P16), _p(ph, P16),
                              low_pow, _p(pa, P16), stride)
        orc.xo_qmf_synthesis(_p(q.copy(), P32), ss, _p(sf, P16), lsb, usb, 6, ctypes.byref(st), low_pow, _p(pb, P16),
                             stride)
        assert np.array_equal(pa, pb), f
        assert np.array_equal(np.array(st.ring[:]), ring) and st.drc_offset == d[0] and st.phase == ph[0], f
