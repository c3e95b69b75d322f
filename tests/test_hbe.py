"""The harmonic transposer's polyphase banks against reference-made chains (tests/golden/hbe_ref.npz,
tools/make_golden_hbe.py: the compiled reference's ixheaacd_real_synth_filt / ixheaacd_complex_anal_filt on its own
transposer instance, delay lines carried, one chain per bank size).  CPU: the oracle (oracle/oracle_hbe.cpp)
reproduces every state CRC.  GPU: xaac_hbe_real_synth_batch / xaac_hbe_cplx_anal_batch through the C ABI with all
chains (and a second, shifted copy of each) as one batch: every state bit-identical to the reference's CRCs and to the
oracle's state on extra random frames."""
import ctypes
import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden_hbe import START_BANDS, FRAMES, chain_input, shift_input, run  # noqa: E402
from hbe_structs import HbeState, new_state  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "hbe_ref.npz"))
PF = ctypes.POINTER(ctypes.c_float)


def test_oracle_matches_reference_chains(oracle):
    crcs, last_time, last_rows = run(oracle.lib, "xo")
    assert np.array_equal(crcs, GOLD["crc"])
    assert np.array_equal(last_time.view(np.uint32), GOLD["last_time"].view(np.uint32))
    assert np.array_equal(last_rows.view(np.uint32), GOLD["last_rows"].view(np.uint32))


def test_parameters_outside_the_tables_are_refused(oracle):
    syn = oracle.lib.xo_hbe_real_synth
    syn.restype, syn.argtypes = ctypes.c_int, [ctypes.POINTER(HbeState), PF, PF, ctypes.c_int]
    z = np.zeros((32, 64), np.float32)
    for size, ks in ((24, 0), (0, 0), (8, -1), (20, 60)):
        st = new_state(9)
        st.synth_size, st.k_start = size, ks
        assert syn(ctypes.byref(st), z.ctypes.data_as(PF), z.ctypes.data_as(PF), 32) == -1


def _states_tensor(torch, dev, states):
    return torch.from_numpy(np.stack([np.frombuffer(bytes(s), np.uint8) for s in states])).to(dev)


@pytest.mark.gpu
def test_gpu_chains_match_reference_and_oracle(oracle):
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    nchain = len(START_BANDS)
    # channels 0..4: the golden chains; 5..9: the same chains one frame late (mixed delay-line contents in one batch);
    # 10: parameters outside the tables
    sbs = START_BANDS + START_BANDS + [9]
    n = len(sbs)
    host = [new_state(sb) for sb in sbs]
    host[-1].synth_size = 24
    state = _states_tensor(torch, dev, host)
    status = torch.full((n,), 7, dtype=torch.int32, device=dev)
    off_in = HbeState.input_buf.offset
    osyn, oana = oracle.lib.xo_hbe_real_synth, oracle.lib.xo_hbe_cplx_anal
    osyn.restype, osyn.argtypes = ctypes.c_int, [ctypes.POINTER(HbeState), PF, PF, ctypes.c_int]
    oana.restype, oana.argtypes = ctypes.c_int, [ctypes.POINTER(HbeState)]
    rng = np.random.default_rng(9)
    for f in range(FRAMES + 3):
        re, im = np.zeros((n, 32, 64), np.float32), np.zeros((n, 32, 64), np.float32)
        for c in range(nchain):
            if f < FRAMES:
                re[c], im[c] = chain_input(c, f)
            else:
                re[c] = (rng.standard_normal((32, 64)) * 2.0 ** rng.integers(-8, 20)).astype(np.float32)
                im[c] = (rng.standard_normal((32, 64)) * 100).astype(np.float32)
            if f >= 1:
                re[nchain + c], im[nchain + c] = chain_input(c, f - 1) if f - 1 < FRAMES else (re[c] * 0.5, im[c] * 2)
        # the apply function's shift of the time signal (hbe_trans.c:235-238), as the host of these two calls does it
        sv = state.view(n, -1)
        fl = sv[:, off_in:off_in + 4 * 1088].contiguous().view(torch.float32).view(n, 1088)
        for c, st in enumerate(host):
            s = st.synth_size
            if s <= 20:
                fl[c, :s] = fl[c, 32 * s:33 * s].clone()
        sv[:, off_in:off_in + 4 * 1088] = fl.view(torch.uint8).view(n, -1)
        ctx.hbe_real_synth_batch(torch.from_numpy(re).to(dev), torch.from_numpy(im).to(dev), state, status)
        ctx.sync()
        got_syn = state.cpu().numpy()
        ctx.hbe_cplx_anal_batch(state, status)
        ctx.sync()
        got_ana = state.cpu().numpy()
        assert status.cpu().tolist() == [0] * (n - 1) + [-1]
        for c in range(n - 1):
            shift_input(host[c])
            assert osyn(ctypes.byref(host[c]), re[c].ctypes.data_as(PF), im[c].ctypes.data_as(PF), 32) == 0
            assert np.array_equal(np.frombuffer(bytes(host[c]), np.uint8), got_syn[c]), ("synthesis", f, c)
            assert oana(ctypes.byref(host[c])) == 0
            d = np.nonzero(np.frombuffer(bytes(host[c]), np.uint8) != got_ana[c])[0]
            assert d.size == 0, ("analysis", f, c, d[:4])
        if f < FRAMES:
            for c in range(nchain):
                assert zlib.crc32(got_syn[c].tobytes()) & 0xffffffff == int(GOLD["crc"][c, f, 0]), (f, c)
                assert zlib.crc32(got_ana[c].tobytes()) & 0xffffffff == int(GOLD["crc"][c, f, 1]), (f, c)
        if f >= 1 and f - 1 < FRAMES:
            for c in range(nchain):
                assert zlib.crc32(got_ana[nchain + c].tobytes()) & 0xffffffff == int(GOLD["crc"][c, f - 1, 1]), (f, c)
    # the refused channel's state is untouched
    assert np.array_equal(got_ana[-1], np.frombuffer(bytes(host[-1]), np.uint8))


from make_golden_hbe import APPLY_FRAMES, APPLY_PITCH, apply_input, state_from_params  # noqa: E402


def _oracle_apply(oracle):
    fn = oracle.lib.xo_hbe_apply
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(HbeState), PF, PF, ctypes.c_int, PF, PF]
    return fn


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def test_oracle_apply_matches_reference_chains(oracle):
    """ixheaacd_qmf_hbe_apply chains made by the reference (state and output-row CRCs per frame): the oracle's
    xo_hbe_apply reproduces all of them from the stored bank parameters"""
    fn = _oracle_apply(oracle)
    for c, par in enumerate(GOLD["apply_params"]):
        st = state_from_params(par)
        for f in range(APPLY_FRAMES):
            re, im = apply_input(c, f)
            pv = np.full((2, 32, 64), 7.5, np.float32)
            assert fn(ctypes.byref(st), re.ctypes.data_as(PF), im.ctypes.data_as(PF), APPLY_PITCH[c], pv[0].ctypes.data_as(PF), pv[1].ctypes.data_as(PF)) == 0
            assert (_crc(bytes(st)), _crc(pv[0]), _crc(pv[1])) == tuple(int(v) for v in GOLD["apply_crc"][c, f]), (c, f)
        assert np.array_equal(pv.view(np.uint32), GOLD["apply_last_pv"][c].view(np.uint32))


@pytest.mark.gpu
def test_gpu_apply_matches_reference_and_oracle(oracle):
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    fn = _oracle_apply(oracle)
    pars = GOLD["apply_params"]
    nchain = len(pars)
    # channels 0..8: the golden chains (the last three with a pitch: cross products); 9..17: the same one frame late,
    # after a silent frame (whose analysis rows hold negative zeros: against the oracle only); 18: a pitch outside its
    # seven bits (refused); 19: a cross-over band outside the row (refused)
    host = [state_from_params(p) for p in pars] + [state_from_params(p) for p in pars] + [state_from_params(pars[1]), state_from_params(pars[1])]
    host[-1].x_over_qmf[1] = 70
    n = len(host)
    pitch_np = np.array(APPLY_PITCH + APPLY_PITCH + [200, 0], np.int32)
    pitch = torch.from_numpy(pitch_np).to(dev)
    state = _states_tensor(torch, dev, host)
    untouched = [bytes(host[-2]), bytes(host[-1])]
    status = torch.full((n,), 7, dtype=torch.int32, device=dev)
    rng = np.random.default_rng(10)
    for f in range(APPLY_FRAMES + 3):
        re, im = np.zeros((n, 32, 64), np.float32), np.zeros((n, 32, 64), np.float32)
        for c in range(nchain):
            if f < APPLY_FRAMES:
                re[c], im[c] = apply_input(c, f)
            else:
                re[c], im[c] = apply_input(c, 0)
                re[c] += (rng.standard_normal((32, 64)) * 2.0 ** rng.integers(-4, 8)).astype(np.float32)
                im[c] *= np.float32(0.5)
            if 1 <= f <= APPLY_FRAMES:
                re[nchain + c], im[nchain + c] = apply_input(c, f - 1)
        re[-2:], im[-2:] = re[:2], im[:2]
        pv_re = torch.full((n, 32, 64), 7.5, dtype=torch.float32, device=dev)
        pv_im = torch.full((n, 32, 64), 7.5, dtype=torch.float32, device=dev)
        ctx.hbe_apply_batch(torch.from_numpy(re).to(dev), torch.from_numpy(im).to(dev), state, pv_re, pv_im, status, pitch)
        ctx.sync()
        got, g_re, g_im = state.cpu().numpy(), pv_re.cpu().numpy(), pv_im.cpu().numpy()
        assert status.cpu().tolist() == [0] * (2 * nchain) + [-1, -1]
        for c in range(2 * nchain):
            pv = np.full((2, 32, 64), 7.5, np.float32)
            assert fn(ctypes.byref(host[c]), re[c].ctypes.data_as(PF), im[c].ctypes.data_as(PF), int(pitch_np[c]), pv[0].ctypes.data_as(PF), pv[1].ctypes.data_as(PF)) == 0
            d = np.nonzero(np.frombuffer(bytes(host[c]), np.uint8) != got[c])[0]
            assert d.size == 0, ("state", f, c, d[:4])
            assert np.array_equal(pv[0].view(np.uint32), g_re[c].view(np.uint32)) and np.array_equal(pv[1].view(np.uint32), g_im[c].view(np.uint32)), ("rows", f, c)
        if f < APPLY_FRAMES:
            for c in range(nchain):
                assert (_crc(got[c]), _crc(g_re[c]), _crc(g_im[c])) == tuple(int(v) for v in GOLD["apply_crc"][c, f]), (f, c)
        assert [bytes(got[-2]), bytes(got[-1])] == untouched
        assert np.all(g_re[-2:] == 7.5) and np.all(g_im[-2:] == 7.5)


def _dft_oracle(oracle):
    from hbe_structs import HbeDftState
    oo = oracle.lib.xo_hbe_dft_anal
    oo.restype = ctypes.c_int
    oo.argtypes = [ctypes.POINTER(HbeDftState), PF, PF, PF, ctypes.c_int, PF, PF]
    return oo, HbeDftState


def test_oracle_dft_bank_matches_reference_chains(oracle):
    """reference-made chains of the DFT transposer's analysis bank (its own coefficient matrices stored with them)"""
    from make_golden_hbe import DFT_FRAMES, dft_input
    oo, S = _dft_oracle(oracle)
    for c, (L, a0) in enumerate(GOLD["dft_sizes"]):
        st = S()
        st.analy_size, st.a_start = int(L), int(a0)
        cre, cim = np.ascontiguousarray(GOLD["dft_coef"][c, 0]), np.ascontiguousarray(GOLD["dft_coef"][c, 1])
        for f in range(DFT_FRAMES):
            t, q = dft_input(c, f)
            assert oo(ctypes.byref(st), t.ctypes.data_as(PF), cre.ctypes.data_as(PF), cim.ctypes.data_as(PF), 32,
                      q[0].ctypes.data_as(PF), q[1].ctypes.data_as(PF)) == 0
            assert (_crc(bytes(st)), _crc(q[0]), _crc(q[1])) == tuple(int(v) for v in GOLD["dft_crc"][c, f]), (c, f)


@pytest.mark.gpu
def test_gpu_dft_bank_matches_reference_chains():
    import torch
    import libxaac_amd
    from hbe_structs import HbeDftState
    from make_golden_hbe import DFT_FRAMES, dft_input
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = len(GOLD["dft_sizes"])
    host = [HbeDftState() for _ in range(n)]
    for c, (L, a0) in enumerate(GOLD["dft_sizes"]):
        host[c].analy_size, host[c].a_start = int(L), int(a0)
    state = _states_tensor(torch, dev, host)
    cre = torch.from_numpy(np.ascontiguousarray(GOLD["dft_coef"][:, 0])).to(dev)
    cim = torch.from_numpy(np.ascontiguousarray(GOLD["dft_coef"][:, 1])).to(dev)
    cfg = torch.arange(n, dtype=torch.int32, device=dev)
    for f in range(DFT_FRAMES):
        ins = [dft_input(c, f) for c in range(n)]
        t = torch.from_numpy(np.stack([i[0] for i in ins])).to(dev)
        qr = torch.from_numpy(np.stack([i[1][0] for i in ins])).to(dev)
        qi = torch.from_numpy(np.stack([i[1][1] for i in ins])).to(dev)
        ctx.hbe_dft_anal_batch(t, cre, cim, state, qr, qi, cfg)
        ctx.sync()
        s, a, b = state.cpu().numpy(), qr.cpu().numpy(), qi.cpu().numpy()
        for c in range(n):
            assert (_crc(s[c]), _crc(a[c]), _crc(b[c])) == tuple(int(v) for v in GOLD["dft_crc"][c, f]), (c, f)


@pytest.mark.gpu
def test_gpu_dft_transposer_analysis_bank(oracle):
    """xaac_hbe_dft_anal_batch_run (ixheaacd_dft_hbe_cplx_anal_filt; the oracle is pinned on the reference for every
    prototype-filter case in tests/test_hbe_oracle_vs_reference.py): a batch with one channel per bank size, three
    configurations' coefficient matrices, chains of frames with the delay lines carried; rows incl. the cleared cells and
    states identical to the oracle's."""
    import torch
    import libxaac_amd
    oo, HbeDftState = _dft_oracle(oracle)
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    cfgs = [(8, 0), (24, 8), (28, 8), (36, 12), (44, 20), (60, 4), (64, 0), (4, 60), (40, 16), (12, 3)]   # (analy_size, a_start)
    n = len(cfgs) + 1
    rng = np.random.default_rng(21)
    coef = np.zeros((n, 2, 64, 128), np.float32)
    for c, (L, a0) in enumerate(cfgs):   # the reference's formula (hbe_dft_trans.c:374-388); any matrices would do here
        k, l = np.arange(64)[:, None], np.arange(128)[None, :]
        ang = np.pi / (2 * L) * ((k + 0.5) * (2 * l - L / 64.0) - L / 64.0 * a0)
        coef[c, 0], coef[c, 1] = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    host = [HbeDftState() for _ in range(n)]
    for c, (L, a0) in enumerate(cfgs):
        host[c].analy_size, host[c].a_start = L, a0
    host[-1].analy_size, host[-1].a_start = 30, 0   # not a multiple of four: refused
    state = _states_tensor(torch, dev, host)
    cre, cim = torch.from_numpy(np.ascontiguousarray(coef[:, 0])).to(dev), torch.from_numpy(np.ascontiguousarray(coef[:, 1])).to(dev)
    cfg = torch.arange(n, dtype=torch.int32, device=dev)
    status = torch.full((n,), 7, dtype=torch.int32, device=dev)
    for frame in range(4):
        t = (rng.standard_normal((n, 2100)) * 2.0 ** rng.integers(-3, 12, (n, 1))).astype(np.float32)
        if frame == 2:
            t[:] = 0
        q = [rng.standard_normal((n, 34, 64)).astype(np.float32) for _ in range(2)]
        g = [torch.from_numpy(a.copy()).to(dev) for a in q]
        ctx.hbe_dft_anal_batch(torch.from_numpy(t).to(dev), cre, cim, state, g[0], g[1], cfg, status)
        ctx.sync()
        got = [a.cpu().numpy() for a in g]
        gs = state.cpu().numpy()
        assert status.cpu().tolist() == [0] * (n - 1) + [-1]
        for c in range(n - 1):
            tt = np.zeros(4096, np.float32)
            tt[:2100] = t[c]
            o = [q[0][c].copy(), q[1][c].copy()]
            assert oo(ctypes.byref(host[c]), tt.ctypes.data_as(PF), coef[c, 0].ctypes.data_as(PF), coef[c, 1].ctypes.data_as(PF), 32,
                      o[0].ctypes.data_as(PF), o[1].ctypes.data_as(PF)) == 0
            for a, b, nm in zip(o, got, ("real", "imag")):
                d = np.argwhere(a.view(np.uint32) != b[c].view(np.uint32))
                assert d.size == 0, (nm, frame, cfgs[c], d[:4].tolist())
            assert np.array_equal(np.frombuffer(bytes(host[c]), np.uint8), gs[c]), ("delay line", frame, cfgs[c])
        assert np.array_equal(got[0][-1], q[0][-1]) and np.array_equal(got[1][-1], q[1][-1])


@pytest.mark.gpu
def test_gpu_apply_with_the_bank_size_hint(oracle):
    """xaac_hbe_apply_batch_desc::max_synth_size = 8 (less LDS per channel in the banks kernel): the same words as without the
    hint for banks of size 4 and 8; a channel with a larger bank is refused with status -1 and left alone"""
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    pars = [[8, 2, 9, 31, 9, 15, 27, 31, 0, 0, 4], [4, 0, 3, 20, 3, 6, 9, 12, 0, 0, 4], [8, 4, 11, 33, 11, 22, 33, 0, 0, 0, 3],
            [12, 6, 15, 41, 15, 29, 41, 0, 0, 0, 3]]
    host = [state_from_params(p) for p in pars]
    n = len(host)
    rng = np.random.default_rng(77)
    a, b = _states_tensor(torch, dev, host), _states_tensor(torch, dev, host)
    big = bytes(host[3])
    for f in range(3):
        re = torch.from_numpy((rng.standard_normal((n, 32, 64)) * 900).astype(np.float32)).to(dev)
        im = torch.from_numpy((rng.standard_normal((n, 32, 64)) * 900).astype(np.float32)).to(dev)
        outs = []
        for st, hint in ((a, 0), (b, 8)):
            pvr, pvi = torch.zeros_like(re), torch.zeros_like(re)
            status = torch.full((n,), 7, dtype=torch.int32, device=dev)
            ctx.hbe_apply_batch(re, im, st, pvr, pvi, status, max_synth_size=hint)
            ctx.sync()
            outs.append((pvr.cpu().numpy(), pvi.cpu().numpy(), status.cpu().numpy()))
        assert outs[0][2].tolist() == [0, 0, 0, 0] and outs[1][2].tolist() == [0, 0, 0, -1]
        for k in range(2):
            assert np.array_equal(outs[0][k][:3].view(np.uint32), outs[1][k][:3].view(np.uint32))
        sa, sb = a.cpu().numpy(), b.cpu().numpy()
        assert np.array_equal(sa[:3], sb[:3]) and sb[3].tobytes() == big
    ctx.close()
