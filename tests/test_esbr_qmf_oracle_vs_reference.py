"""eSBR ("Path A", the reference's default -esbr:1) QMF banks: the oracle's ring-faithful restatement
(oracle/oracle_qmf.cpp: xo_esbr_analysis / xo_esbr_synthesis, slot transforms from libxaac_amd/csrc/sbr_qmf.h on the
32-bit constants) against the compiled reference's own ixheaacd_esbr_analysis_filt_block (sbr_dec.c:185) and the bank
loop of ixheaacd_esbr_synthesis_filt_block (sbr_dec.c:447), driven through oracle/ref_sbr_adapter.c: float outputs
bit-identical (compared as raw words), WORD32 rings and positions identical, over chains of frames with the state carried,
at several signal levels including full scale."""
import ctypes

import numpy as np
import pytest

PF = ctypes.POINTER(ctypes.c_float)
P32 = ctypes.POINTER(ctypes.c_int32)


def _bind(lib, prefix):
    a = getattr(lib, prefix + "_esbr_analysis")
    s = getattr(lib, prefix + "_esbr_synthesis")
    a.restype = s.restype = None
    a.argtypes = [PF, P32, P32, P32, PF, PF]
    s.argtypes = [PF, PF, P32, P32, P32, PF]
    return a, s


def ana(fn, core, ring, pos, win):
    re, im = np.zeros((32, 64), np.float32), np.zeros((32, 64), np.float32)
    p, w = ctypes.c_int32(pos), ctypes.c_int32(win)
    fn(core.ctypes.data_as(PF), ring.ctypes.data_as(P32), ctypes.byref(p), ctypes.byref(w), re.ctypes.data_as(PF),
       im.ctypes.data_as(PF))
    return re, im, p.value, w.value


def syn(fn, re, im, ring, drc, filt):
    out = np.zeros(2048, np.float32)
    d, f = ctypes.c_int32(drc), ctypes.c_int32(filt)
    fn(re.ctypes.data_as(PF), im.ctypes.data_as(PF), ring.ctypes.data_as(P32), ctypes.byref(d), ctypes.byref(f),
       out.ctypes.data_as(PF))
    return out, d.value, f.value


@pytest.mark.parametrize("amp", [1.0, 0.05, 1e-4, 0.999, 2e6])  # the last: (WORD32) of values outside int32
def test_analysis_chain(oracle, reference, amp):
    ra, _ = _bind(reference.lib, "ref")
    oa, _ = _bind(oracle.lib, "xo")
    rng = np.random.default_rng(int(amp * 1e6) + 1)
    ring_r, ring_o = np.zeros(320, np.int32), np.zeros(320, np.int32)
    pr = wr = po = wo = 0
    for f in range(7):
        core = np.ascontiguousarray((rng.uniform(-1, 1, 1024) * amp).astype(np.float32))
        if f == 3:
            core[::7] = np.float32(amp)          # plateaus / sign flips
            core[1::7] = np.float32(-amp)
        rr, ri, pr, wr = ana(ra, core, ring_r, pr, wr)
        xr, xi, po, wo = ana(oa, core, ring_o, po, wo)
        assert (pr, wr) == (po, wo), f
        assert np.array_equal(ring_r, ring_o), f
        assert np.array_equal(rr.view(np.uint32), xr.view(np.uint32)), (f, int(np.sum(rr != xr)))
        assert np.array_equal(ri.view(np.uint32), xi.view(np.uint32)), f
        assert np.any(rr[:, :32] != 0) and not np.any(rr[:, 32:])


@pytest.mark.parametrize("amp", [1.0, 30.0, 1e-3, 4000.0, 3e8])  # the last: (WORD32) of values outside int32
def test_synthesis_chain(oracle, reference, amp):
    _, rs = _bind(reference.lib, "ref")
    _, os_ = _bind(oracle.lib, "xo")
    rng = np.random.default_rng(int(amp * 1e3) + 5)
    ring_r, ring_o = np.zeros(1280, np.int32), np.zeros(1280, np.int32)
    dr = fr = do = fo = 0
    for f in range(7):
        re = np.ascontiguousarray((rng.standard_normal((32, 64)) * amp).astype(np.float32))
        im = np.ascontiguousarray((rng.standard_normal((32, 64)) * amp).astype(np.float32))
        re[:, 40:] = 0
        im[:, 40:] = 0
        outr, dr, fr = syn(rs, re, im, ring_r, dr, fr)
        outo, do, fo = syn(os_, re, im, ring_o, do, fo)
        assert (dr, fr) == (do, fo), f
        assert np.array_equal(ring_r, ring_o), f
        assert np.array_equal(outr.view(np.uint32), outo.view(np.uint32)), (f, int(np.sum(outr != outo)))


def test_analysis_into_synthesis_round_trip(oracle, reference):
    """both banks back to back on a sine (the bank pair is near-perfect-reconstruction with a delay): oracle = reference
    bit for bit, and the reconstruction is the delayed input within the prototype filter's ripple"""
    ra, rs = _bind(reference.lib, "ref")
    oa, os_ = _bind(oracle.lib, "xo")
    n = np.arange(8 * 1024)
    x = (0.5 * np.sin(2 * np.pi * 997.0 * n / 24000.0)).astype(np.float32)
    st = {k: dict(ar=np.zeros(320, np.int32), sr=np.zeros(1280, np.int32), p=0, w=0, d=0, f=0) for k in ("r", "o")}
    outs = {"r": [], "o": []}
    for fidx in range(8):
        core = np.ascontiguousarray(x[1024 * fidx:1024 * (fidx + 1)])
        for k, (fa, fs) in (("r", (ra, rs)), ("o", (oa, os_))):
            s = st[k]
            re, im, s["p"], s["w"] = ana(fa, core, s["ar"], s["p"], s["w"])
            re2, im2 = np.zeros((32, 64), np.float32), np.zeros((32, 64), np.float32)
            re2[:, :32], im2[:, :32] = re[:, :32], im[:, :32]      # 32 analysed bands into the 64-band bank: 2x upsampling
            y, s["d"], s["f"] = syn(fs, re2, im2, s["sr"], s["d"], s["f"])
            outs[k].append(y)
    yr, yo = np.concatenate(outs["r"]), np.concatenate(outs["o"])
    assert np.array_equal(yr.view(np.uint32), yo.view(np.uint32))
    assert np.abs(yo).max() > 0.1


def _bind_nb(lib, prefix):
    a = getattr(lib, prefix + "_esbr_analysis_nb")
    a.restype = None
    a.argtypes = [PF, ctypes.c_int, ctypes.c_int, P32, P32, P32, PF, PF]
    return a


def ana_nb(fn, core, nb, n_slots, ring, pos, win):
    re, im = np.zeros((n_slots, 64), np.float32), np.zeros((n_slots, 64), np.float32)
    p, w = ctypes.c_int32(pos), ctypes.c_int32(win)
    fn(core.ctypes.data_as(PF), nb, n_slots, ring.ctypes.data_as(P32), ctypes.byref(p), ctypes.byref(w),
       re.ctypes.data_as(PF), im.ctypes.data_as(PF))
    return re, im, p.value, w.value


@pytest.mark.parametrize("nb,n_slots", [(24, 32), (16, 64), (32, 32), (24, 30), (16, 60)])
@pytest.mark.parametrize("amp", [1.0, 0.05, 1e-4, 0.999, 2e6])
def test_analysis_chain_of_the_8_3_and_4_1_banks(oracle, reference, nb, n_slots, amp):
    """The 24- and 16-channel banks (sbr_dec.c:213-236; the general FFT's 12- and 8-point forward transforms inside
    ixheaacd_esbr_cos_sin_mod, generic:1317-1369) against the reference's own function, state carried over 13 frames (the
    10-block ring and the window pointers wrap several times)."""
    ra, oa = _bind_nb(reference.lib, "ref"), _bind_nb(oracle.lib, "xo")
    rng = np.random.default_rng(int(amp * 1e6) + nb)
    ring_r, ring_o = np.zeros(320, np.int32), np.zeros(320, np.int32)
    pr = wr = po = wo = 0
    for f in range(13):
        core = np.ascontiguousarray((rng.uniform(-1, 1, nb * n_slots) * amp).astype(np.float32))
        if f == 3:
            core[::7] = np.float32(amp)
            core[1::7] = np.float32(-amp)
        rr, ri, pr, wr = ana_nb(ra, core, nb, n_slots, ring_r, pr, wr)
        xr, xi, po, wo = ana_nb(oa, core, nb, n_slots, ring_o, po, wo)
        assert (pr, wr) == (po, wo), f
        assert np.array_equal(ring_r, ring_o), f
        assert np.array_equal(rr.view(np.uint32), xr.view(np.uint32)), (f, int(np.sum(rr != xr)))
        assert np.array_equal(ri.view(np.uint32), xi.view(np.uint32)), f
        assert (amp < 0.01 or np.any(rr[:, :nb] != 0)) and not np.any(rr[:, nb:])


@pytest.mark.parametrize("amp", [1.0, 30.0, 1e-3, 4000.0, 3e8])
def test_down_sampled_synthesis_chain(oracle, reference, amp):
    """the 32-channel synthesis bank of the eSBR branch (-dsample: sbr_dec.c:605-628) against the reference's own function, state
    carried over 13 frames (ring and window positions wrap)"""
    fr_, fo_ = reference.lib.ref_esbr_synthesis_ds, oracle.lib.xo_esbr_synthesis_ds
    for fn in (fr_, fo_):
        fn.restype = None
        fn.argtypes = [PF, PF, P32, P32, P32, PF]
    rng = np.random.default_rng(int(amp * 1e3) + 9)
    ring_r, ring_o = np.zeros(1280, np.int32), np.zeros(1280, np.int32)
    dr = fr = do = fo = 0
    for f in range(13):
        re = np.ascontiguousarray((rng.standard_normal((32, 64)) * amp).astype(np.float32))
        im = np.ascontiguousarray((rng.standard_normal((32, 64)) * amp).astype(np.float32))
        outr, dr, fr = syn(fr_, re, im, ring_r, dr, fr)
        outo, do, fo = syn(fo_, re, im, ring_o, do, fo)
        assert (dr, fr) == (do, fo), f
        assert np.array_equal(ring_r, ring_o), f
        assert np.array_equal(outr[:1024].view(np.uint32), outo[:1024].view(np.uint32)), (f, int(np.sum(outr[:1024] != outo[:1024])))
        assert (amp < 0.01 or np.any(outr[:1024] != 0)) and not np.any(outo[1024:])
