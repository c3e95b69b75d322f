"""USAC FD IMDCT against reference-made chains (tests/golden/usac_imdct_ref.npz, tools/make_golden_usac_imdct.py: the
compiled reference's ixheaacd_fd_frm_dec with the overlap carried along window-sequence walks, all five sequences, both
shapes, levels from silence to full scale).  CPU: the oracle reproduces every CRC.  GPU: xaac_usac_imdct_process_batch
through the C ABI, all chains as one batch per frame step; a large random batch against the oracle; malformed side info
is refused per channel-frame with the neighbours bit-exact."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden_usac_imdct import CHAINS, CHAINS_768, FRAMES, FRAMES_768, chain_coef, crc  # noqa: E402
import test_usac_oracle_vs_reference as t  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "usac_imdct_ref.npz"))


def _start_shape_prev(c):
    return c & 1


def _fixture(ccfl):
    return (CHAINS, FRAMES, "") if ccfl == 1024 else (CHAINS_768, FRAMES_768, "768")


@pytest.mark.parametrize("ccfl", [1024, 768])
def test_oracle_matches_reference_chains(oracle, ccfl):
    chains, frames, key = _fixture(ccfl)
    assert chains * frames >= 500
    for c in range(chains):
        ov = np.zeros(ccfl, np.int32)
        shape_prev = _start_shape_prev(c)
        for f in range(frames):
            seq, shape = (int(v) for v in GOLD["side" + key][c, f])
            rc, _, ov, out = t.orc_call(oracle, chain_coef(c, f, ccfl), ov, seq, shape, shape_prev)
            assert rc == 0
            assert (crc(out), crc(ov)) == tuple(int(v) for v in GOLD["crc" + key][c, f]), (c, f, seq)
            shape_prev = shape
        assert np.array_equal(out, GOLD["last" + key][c, 0]) and np.array_equal(ov, GOLD["last" + key][c, 1])


@pytest.mark.gpu
@pytest.mark.parametrize("ccfl", [1024, 768])
def test_gpu_reference_chains(ccfl):
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    chains, frames, key = _fixture(ccfl)
    n = chains + 1  # + a copy of chain 0: the batch is not a multiple of the workgroup's four channel-frames
    idx = list(range(chains)) + [0]
    ov = torch.zeros((n, ccfl), dtype=torch.int32, device=dev)
    sp = torch.tensor([_start_shape_prev(c) for c in idx], dtype=torch.uint8, device=dev)
    out = torch.zeros((n, ccfl), dtype=torch.int32, device=dev)
    tm = torch.zeros((n, ccfl), dtype=torch.float32, device=dev)
    status = torch.full((n,), 7, dtype=torch.int32, device=dev)
    for f in range(frames):
        coef = torch.from_numpy(np.stack([chain_coef(c, f, ccfl) for c in idx])).to(dev)
        ics = torch.from_numpy(np.stack([GOLD["side" + key][c, f] for c in idx])).to(dev)
        ctx.usac_imdct_process_batch(coef, ics, ov, sp, out, tm, status, ccfl=ccfl)
        ctx.sync()
        o, v = out.cpu().numpy(), ov.cpu().numpy()
        assert not status.cpu().numpy().any()
        for c in range(chains):
            assert (crc(o[c]), crc(v[c])) == tuple(int(x) for x in GOLD["crc" + key][c, f]), (c, f, GOLD["side" + key][c, f])
        assert np.array_equal(o[chains], o[0]) and np.array_equal(v[chains], v[0])
        assert np.array_equal(tm.cpu().numpy(), o.astype(np.float32) * np.float32(2.0 ** -15))
        assert np.array_equal(sp.cpu().numpy(), ics.cpu().numpy()[:, 1])
    assert np.array_equal(o[:chains], GOLD["last" + key][:, 0]) and np.array_equal(v[:chains], GOLD["last" + key][:, 1])


@pytest.mark.gpu
@pytest.mark.parametrize("L", [1024, 768])
def test_gpu_large_batch_vs_oracle(oracle, L):
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = 4099
    rng = np.random.default_rng(3)
    ov_h = np.zeros((n, L), np.int32)
    sp_h = rng.integers(0, 2, n).astype(np.uint8)
    seq = rng.integers(0, 5, n)
    ov = torch.from_numpy(ov_h).to(dev)
    sp = torch.from_numpy(sp_h).to(dev)
    out = torch.zeros((n, L), dtype=torch.int32, device=dev)
    check = np.concatenate([np.arange(0, 40), rng.integers(0, n, 60), [n - 3, n - 2, n - 1]])
    for f in range(3):
        lvl = rng.integers(0, 27, (n, 1))
        coef = (rng.integers(-2 ** 31, 2 ** 31, (n, L)) >> lvl).astype(np.int32)
        coef[rng.integers(0, n, 50)] = 0
        shape = rng.integers(0, 2, n).astype(np.uint8)
        ics = np.stack([seq.astype(np.uint8), shape], 1)
        ctx.usac_imdct_process_batch(torch.from_numpy(coef).to(dev), torch.from_numpy(ics).to(dev), ov, sp, out, ccfl=L)
        ctx.sync()
        o, v = out.cpu().numpy(), ov.cpu().numpy()
        for c in check:
            rc, _, nov, xo = t.orc_call(oracle, coef[c], ov_h[c], int(seq[c]), int(shape[c]), int(sp_h[c]))
            assert np.array_equal(xo, o[c]), (f, c, int(seq[c]))
            assert np.array_equal(nov, v[c]), (f, c, int(seq[c]))
        ov_h, sp_h = v.copy(), shape
        seq = np.array([rng.choice(t.NEXT[int(s)]) for s in seq])
    assert np.any(o)


@pytest.mark.gpu
def test_gpu_malformed_side_info_is_refused(oracle):
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = 11
    rng = np.random.default_rng(9)
    coef = (rng.integers(-2 ** 31, 2 ** 31, (n, 1024)) >> 6).astype(np.int32)
    ov_h = (rng.integers(-2 ** 31, 2 ** 31, (n, 1024)) >> 18).astype(np.int32)
    ics = np.stack([rng.integers(0, 5, n), rng.integers(0, 2, n)], 1).astype(np.uint8)
    sp_h = rng.integers(0, 2, n).astype(np.uint8)
    ics[2, 0] = 5
    ics[5, 1] = 2
    sp_h[8] = 200
    ov, sp = torch.from_numpy(ov_h).to(dev), torch.from_numpy(sp_h).to(dev)
    out = torch.full((n, 1024), 77, dtype=torch.int32, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.usac_imdct_process_batch(torch.from_numpy(coef).to(dev), torch.from_numpy(ics).to(dev), ov, sp, out, None, status)
    ctx.sync()
    st, o, v, s2 = status.cpu().numpy(), out.cpu().numpy(), ov.cpu().numpy(), sp.cpu().numpy()
    for c in range(n):
        if c in (2, 5, 8):
            assert st[c] == libxaac_amd.BAD_WINDOW_SEQ
            assert np.all(o[c] == 77) and np.array_equal(v[c], ov_h[c]) and s2[c] == sp_h[c]
        else:
            rc, _, nov, xo = t.orc_call(oracle, coef[c], ov_h[c], int(ics[c, 0]), int(ics[c, 1]), int(sp_h[c]))
            assert st[c] == 0 and np.array_equal(xo, o[c]) and np.array_equal(nov, v[c]) and s2[c] == ics[c, 1]


# ---- LPD -> FD transitions: tests/golden/usac_lpd_ref.npz (tools/make_golden_usac_imdct.py: make_lpd) -------------------------
from make_golden_usac_imdct import LPD_CHAINS, LPD_FRAMES  # noqa: E402

LPD = np.load(os.path.join(ROOT, "tests", "golden", "usac_lpd_ref.npz"))


@pytest.mark.parametrize("ccfl", [1024, 768])
def test_oracle_matches_reference_lpd_chains(oracle, ccfl):
    """544 reference-made frames per frame length, a third of them behind an LPD frame (slope of 2 lfac samples, Q15 -> float ->
    Q15 around the LPD decoder's post filter), two in three of those with the reference's own FAC signal"""
    k = str(ccfl)
    assert LPD["side" + k][:, :, 2].sum() > 150 and LPD["side" + k][:, :, 3].sum() > 100
    for c in range(LPD_CHAINS):
        ov = np.zeros(ccfl, np.int32)
        shape_prev = c & 1
        for f in range(LPD_FRAMES):
            seq, shape, td, fac = (int(v) for v in LPD["side" + k][c, f])
            if td:
                shape_prev = 0
            sig = np.ascontiguousarray(LPD["fac" + k][c, f]) if fac else None
            rc, _, ov, out = t.orc_call_lpd(oracle, chain_coef(c + 32, f, ccfl), ov, seq, shape, shape_prev, td, sig, int(LPD["fac_q" + k][c, f]))
            assert rc == 0
            assert (crc(out), crc(ov)) == tuple(int(v) for v in LPD["crc" + k][c, f]), (c, f, seq, td, fac)
            shape_prev = shape


@pytest.mark.gpu
@pytest.mark.parametrize("ccfl", [1024, 768])
def test_gpu_reference_lpd_chains(ccfl):
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    k = str(ccfl)
    n = LPD_CHAINS
    ov = torch.zeros((n, ccfl), dtype=torch.int32, device=dev)
    sp = torch.tensor([c & 1 for c in range(n)], dtype=torch.uint8, device=dev)
    out = torch.zeros((n, ccfl), dtype=torch.int32, device=dev)
    tm = torch.zeros((n, ccfl), dtype=torch.float32, device=dev)
    status = torch.full((n,), 7, dtype=torch.int32, device=dev)
    for f in range(LPD_FRAMES):
        side = LPD["side" + k][:, f]
        coef = torch.from_numpy(np.stack([chain_coef(c + 32, f, ccfl) for c in range(n)])).to(dev)
        ics = torch.from_numpy(np.ascontiguousarray(side[:, :2])).to(dev)
        flags = torch.from_numpy((side[:, 2] | (side[:, 3] << 1)).astype(np.uint8)).to(dev)
        fac = torch.from_numpy(np.concatenate([LPD["fac_q" + k][:, f, None], LPD["fac" + k][:, f]], 1).astype(np.int32)).to(dev)
        sp[torch.from_numpy(side[:, 2] != 0).to(dev)] = 0      # the shape an LPD frame leaves behind
        ctx.usac_imdct_process_batch(coef, ics, ov, sp, out, tm, status, ccfl=ccfl, lpd_flags=flags, fac=fac)
        ctx.sync()
        o, v = out.cpu().numpy(), ov.cpu().numpy()
        assert not status.cpu().numpy().any()
        for c in range(n):
            assert (crc(o[c]), crc(v[c])) == tuple(int(x) for x in LPD["crc" + k][c, f]), (c, f, side[c])
        assert np.array_equal(tm.cpu().numpy(), o.astype(np.float32) * np.float32(2.0 ** -15))


def lpd_sides():
    """the LPD-side inputs tools/make_golden_usac_imdct.py: make_lpd drew for every frame (its generator replayed in its order):
    {ccfl: int32 [chains, frames, 402] as struct xaac_usac_fac_in}"""
    rng = np.random.default_rng(77)
    sides = {}
    for ccfl in (1024, 768):
        a = np.zeros((LPD_CHAINS, LPD_FRAMES, 129 + 17 + 256), np.int32)
        for c in range(LPD_CHAINS):
            for f in range(LPD_FRAMES):
                seq, _, _, fac = (int(v) for v in LPD["side" + str(ccfl)][c, f])
                fd, lpc, zir = t.lpd_side(rng, ccfl, seq, fac)
                a[c, f, :129] = fd
                a[c, f, 129:146] = lpc.view(np.int32)
                a[c, f, 146:] = zir[:256].view(np.int32)
        sides[ccfl] = a
    return sides


@pytest.mark.parametrize("ccfl", [1024, 768])
def test_oracle_makes_the_fac_signals_of_the_lpd_chains(oracle, ccfl):
    """ixheaacd_cal_fac_data restated (usac_fac.h) on the inputs the fixture's maker drew: the reference-made signals and exponents"""
    fn = oracle.lib.xo_usac_cal_fac
    fn.restype = ctypes.c_int
    PF = ctypes.POINTER(ctypes.c_float)
    P32 = ctypes.POINTER(ctypes.c_int32)
    fn.argtypes = [ctypes.c_int] * 3 + [P32, PF, PF, P32, P32]
    k, sides, n = str(ccfl), lpd_sides()[ccfl], 0
    for c in range(LPD_CHAINS):
        for f in range(LPD_FRAMES):
            seq, _, td, fac = (int(v) for v in LPD["side" + k][c, f])
            if not fac:
                continue
            s = sides[c, f]
            fd, lpc, zir = np.ascontiguousarray(s[:129]), np.ascontiguousarray(s[129:146]).view(np.float32), np.ascontiguousarray(s[146:]).view(np.float32)
            out, q = np.zeros(256, np.int32), np.zeros(1, np.int32)
            assert fn(ccfl, seq, td, fd.ctypes.data_as(P32), lpc.ctypes.data_as(PF), zir.ctypes.data_as(PF), out.ctypes.data_as(P32), q.ctypes.data_as(P32)) == 0
            lfac = ccfl >> 4 if seq == 2 else ccfl >> 3
            assert int(q[0]) == int(LPD["fac_q" + k][c, f]) and np.array_equal(out[:2 * lfac], LPD["fac" + k][c, f][:2 * lfac]), (c, f)
            n += 1
    assert n > 50


@pytest.mark.gpu
@pytest.mark.parametrize("ccfl", [1024, 768])
def test_gpu_reference_lpd_chains_with_the_fac_signal_made_on_the_device(ccfl):
    """the same reference-made walks with xaac_usac_imdct_batch.fac_in: ixheaacd_cal_fac_data runs on the device (xaac_usac_fac_kernel)
    on the LPD-side inputs and the frame takes its signal from there -- the signals equal the reference's own, word and exponent,
    and so do output and overlap of every frame"""
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    k, n, sides = str(ccfl), LPD_CHAINS, lpd_sides()[ccfl]
    ov = torch.zeros((n, ccfl), dtype=torch.int32, device=dev)
    sp = torch.tensor([c & 1 for c in range(n)], dtype=torch.uint8, device=dev)
    out = torch.zeros((n, ccfl), dtype=torch.int32, device=dev)
    status = torch.full((n,), 7, dtype=torch.int32, device=dev)
    work = torch.full((n, 257), 5, dtype=torch.int32, device=dev)
    n_fac = 0
    for f in range(LPD_FRAMES):
        side = LPD["side" + k][:, f]
        coef = torch.from_numpy(np.stack([chain_coef(c + 32, f, ccfl) for c in range(n)])).to(dev)
        ics = torch.from_numpy(np.ascontiguousarray(side[:, :2])).to(dev)
        flags = torch.from_numpy((side[:, 2] | (side[:, 3] << 1)).astype(np.uint8)).to(dev)
        fin = torch.from_numpy(np.ascontiguousarray(sides[:, f])).to(dev)
        sp[torch.from_numpy(side[:, 2] != 0).to(dev)] = 0
        ctx.usac_imdct_process_batch(coef, ics, ov, sp, out, None, status, ccfl=ccfl, lpd_flags=flags, fac_in=fin, fac_work=work)
        ctx.sync()
        o, v, w = out.cpu().numpy(), ov.cpu().numpy(), work.cpu().numpy()
        assert not status.cpu().numpy().any()
        for c in range(n):
            assert (crc(o[c]), crc(v[c])) == tuple(int(x) for x in LPD["crc" + k][c, f]), (c, f, side[c])
            if side[c, 3]:
                lfac = ccfl >> 4 if side[c, 0] == 2 else ccfl >> 3
                assert w[c, 0] == LPD["fac_q" + k][c, f] and np.array_equal(w[c, 1:1 + 2 * lfac], LPD["fac" + k][c, f][:2 * lfac]), (c, f)
                n_fac += 1
    assert n_fac > 50


@pytest.mark.gpu
def test_gpu_fac_without_lpd_frame_is_refused(oracle):
    import ctypes
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = 8
    rng = np.random.default_rng(4)
    coef_h = (rng.integers(-2 ** 31, 2 ** 31, (n, 1024)) >> 8).astype(np.int32)
    coef = torch.from_numpy(coef_h).to(dev)
    ov_h = (rng.integers(-2 ** 31, 2 ** 31, (n, 1024)) >> 18).astype(np.int32)
    ov, sp = torch.from_numpy(ov_h).to(dev), torch.tensor([0, 0, 1, 0, 0, 0, 0, 0], dtype=torch.uint8, device=dev)
    ics_h = np.array([[3, 0], [3, 0], [3, 0], [2, 1], [0, 0], [3, 0], [2, 0], [4, 1]], np.uint8)
    ics = torch.from_numpy(ics_h).to(dev)
    # FAC without LPD; fine; LPD + KBD 256: no window; fine; fine; then three frames whose fac.q would make a shift count leave 0..31
    flags = torch.tensor([2, 3, 1, 1, 0, 3, 3, 3], dtype=torch.uint8, device=dev)
    fac_h = np.zeros((n, 257), np.int32)
    fac_h[5, 0], fac_h[6, 0], fac_h[7, 0] = 60, -40, -18
    fac = torch.from_numpy(fac_h).to(dev)
    out = torch.full((n, 1024), 77, dtype=torch.int32, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.usac_imdct_process_batch(coef, ics, ov, sp, out, None, status, lpd_flags=flags, fac=fac)
    ctx.sync()
    st, o, v = status.cpu().numpy(), out.cpu().numpy(), ov.cpu().numpy()
    assert st[0] < 0 and st[0] != libxaac_amd.BAD_WINDOW_SEQ      # XAAC_FATAL_BAD_ARG
    assert st[2] == libxaac_amd.BAD_WINDOW_SEQ and list(st[[1, 3, 4]]) == [0, 0, 0]
    assert list(st[5:8]) == [st[0]] * 3                            # fac.q outside what the windowing can shift by
    for c in (0, 2, 5, 6, 7):
        assert np.all(o[c] == 77) and np.array_equal(v[c], ov_h[c])
    assert np.any(o[1] != 77) and np.any(o[3] != 77)
    # the oracle refuses the same three
    P32 = ctypes.POINTER(ctypes.c_int32)
    for c in (5, 6, 7):
        cf, vv, oo = coef_h[c].copy(), ov_h[c].copy(), np.zeros(1024, np.int32)
        fd = np.ascontiguousarray(fac_h[c, 1:])
        assert oracle.lib.xo_usac_fd_imdct_lpd(cf.ctypes.data_as(P32), vv.ctypes.data_as(P32), 1024, int(ics_h[c, 0]), int(ics_h[c, 1]), 0, 1,
                                               fd.ctypes.data_as(P32), int(fac_h[c, 0]), oo.ctypes.data_as(P32)) == -1
        assert np.array_equal(vv, ov_h[c])
