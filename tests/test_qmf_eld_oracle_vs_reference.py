"""The LD / ELD flavour of the complex QMF analysis bank (ixheaacd_cplx_anal_qmffilt with AOT_ER_AAC_ELD: low-delay
prototype qmf_c_eld3, ixheaacd_sbr_qmfanal32_winadd_eld qmf_dec.c:484-535, ELD post-modulation twiddles): the oracle's
literal restatement (oracle/oracle_qmf.cpp: xo_qmf_analysis_eld) against the compiled reference, chains of frames of 16
and of 15 slots with ring and pointer state carried, full-scale and quiet input."""
import ctypes

import numpy as np
import pytest

P16 = ctypes.POINTER(ctypes.c_int16)
P32 = ctypes.POINTER(ctypes.c_int32)


def bind(lib, name):
    fn = getattr(lib, name)
    fn.restype = None
    fn.argtypes = [P16, ctypes.c_int, P16, P16, ctypes.c_int, ctypes.c_int, P32, ctypes.c_int]
    return fn


@pytest.mark.parametrize("n_slots", [16, 15])
def test_eld_analysis_chain(oracle, reference, n_slots):
    rf, of = bind(reference.lib, "ref_qmf_analysis_eld"), bind(oracle.lib, "xo_qmf_analysis_eld")
    rng = np.random.default_rng(40 + n_slots)
    ring_r, ring_o = np.zeros(320, np.int16), np.zeros(320, np.int16)
    st_r, st_o = np.array([0, 0, 32, 0], np.int16), np.array([0, 0, 32, 0], np.int16)
    for frame in range(23):
        amp = [32767, 3000, 12, 32767][frame % 4]
        pcm = rng.integers(-amp, amp + 1, 32 * n_slots).astype(np.int16)
        if frame % 7 == 3:
            pcm[:] = 32767 if frame % 2 else -32768
        usb = int(rng.integers(0, 33))
        qr, qo = np.full((n_slots, 128), 5, np.int32), np.full((n_slots, 128), 5, np.int32)
        for fn, ring, st, q in ((rf, ring_r, st_r, qr), (of, ring_o, st_o, qo)):
            fn(pcm.ctypes.data_as(P16), 1, ring.ctypes.data_as(P16), st.ctypes.data_as(P16), n_slots, usb, q.ctypes.data_as(P32), 128)
        assert np.array_equal(qr, qo), frame
        assert np.array_equal(ring_r, ring_o) and np.array_equal(st_r, st_o), (frame, st_r, st_o)
    assert st_r[0] != 0 or n_slots == 16


def bind_syn(lib, name):
    fn = getattr(lib, name)
    fn.restype = None
    fn.argtypes = [P32, ctypes.c_int, P16, ctypes.c_int, ctypes.c_int, ctypes.c_int, P16, P16, ctypes.c_int, P16, ctypes.c_int]
    return fn


@pytest.mark.parametrize("n_slots", [16, 15])
def test_eld_synthesis_chain(oracle, reference, n_slots):
    """ixheaacd_cplx_synt_qmffilt with AOT_ER_AAC_ELD (pre-twiddle, 64-channel inverse modulation, the ELD rounding
    routine, 10-tap window-add on qmf_c_eld with output shift 2, fp / sixty4 carried between frames): PCM, ring and the
    four state words identical over chains of frames, region scales varied, levels up to clipping"""
    rf, of = bind_syn(reference.lib, "ref_qmf_synthesis_eld"), bind_syn(oracle.lib, "xo_qmf_synthesis_eld")
    rng = np.random.default_rng(70 + n_slots)
    ring_r, ring_o = np.zeros(1280, np.int16), np.zeros(1280, np.int16)
    st_r, st_o = np.array([0, 0, 0, 64], np.int16), np.array([0, 0, 0, 64], np.int16)
    for frame in range(21):
        level = 2.0 ** rng.integers(8, 30)
        q = (rng.standard_normal((n_slots, 128)) * level).clip(-2 ** 31, 2 ** 31 - 1).astype(np.int32)
        if frame % 6 == 5:
            q[:] = 2 ** 31 - 1 if frame % 2 else -2 ** 31
        sf = np.array([rng.integers(-12, 4), rng.integers(-12, 4), rng.integers(-12, 4), rng.integers(-10, 2)], np.int16)
        lsb = int(rng.integers(0, 40))
        usb = int(rng.integers(lsb, 65))
        split = int(rng.integers(0, n_slots + 1))
        pr, po = np.zeros(64 * n_slots, np.int16), np.zeros(64 * n_slots, np.int16)
        left = {}
        for fn, ring, st, pcm in ((rf, ring_r, st_r, pr), (of, ring_o, st_o, po)):
            qq = q.copy()
            fn(qq.ctypes.data_as(P32), 128, sf.ctypes.data_as(P16), lsb, usb, split, ring.ctypes.data_as(P16), st.ctypes.data_as(P16),
               n_slots, pcm.ctypes.data_as(P16), 1)
            left[fn is rf] = qq
        # the reference rescales its input in place, hands the rescaled rows on through qmf_real_out / qmf_imag_out and then uses
        # the input rows as work space; the restatement's region scale alone reproduces what is handed on
        handed = np.zeros_like(q)
        ho = reference.lib.ref_qmf_synthesis_eld_handed_on
        ho.restype = None
        ho.argtypes = [P32, ctypes.c_int, ctypes.c_int]
        ho(handed.ctypes.data_as(P32), n_slots, 128)
        scaled = np.zeros_like(q)
        rs = oracle.lib.xo_qmf_eld_region_scale
        rs.restype = None
        rs.argtypes = [P32, ctypes.c_int, P16, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P32]
        rs(q.ctypes.data_as(P32), 128, sf.ctypes.data_as(P16), lsb, usb, split, n_slots, scaled.ctypes.data_as(P32))
        assert np.array_equal(handed, scaled) and np.array_equal(left[False], q), frame
        assert np.array_equal(pr, po), (frame, np.nonzero(pr != po)[0][:5])
        assert np.array_equal(ring_r, ring_o) and np.array_equal(st_r, st_o), (frame, st_r, st_o)
