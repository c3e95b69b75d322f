"""eSBR (Path A) QMF banks against reference-made chains (tests/golden/esbr_qmf_ref.npz, tools/make_golden_esbr_qmf.py:
the compiled reference's ixheaacd_esbr_analysis_filt_block / ixheaacd_esbr_synthesis_filt_block with the state carried).
CPU: the oracle (oracle/oracle_qmf.cpp) reproduces every CRC.  GPU: xaac_esbr_qmf_analysis_batch /
xaac_esbr_qmf_synthesis_batch through the C ABI, all chains as one batch (odd channel count: the last wave is half
empty), float words / rings / positions identical to the reference's and to the oracle's on extra random frames."""
import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden_esbr_qmf import CHAINS, FRAMES, NB_BANKS, chain_input, chain_input_nb  # noqa: E402
import test_esbr_qmf_oracle_vs_reference as t  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "esbr_qmf_ref.npz"))


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def test_oracle_matches_reference_chains(oracle):
    oa, os_ = t._bind(oracle.lib, "xo")
    for c in range(CHAINS):
        ring, pos, win = np.zeros(320, np.int32), 0, 0
        for f in range(FRAMES):
            re, im, pos, win = t.ana(oa, chain_input(0, c, f), ring, pos, win)
            assert (crc(re), crc(im), crc(ring), (pos << 16) | win) == tuple(int(x) for x in GOLD["ana_crc"][c, f]), (c, f)
        assert np.array_equal(re.view(np.uint32), GOLD["ana_last"][c, 0].view(np.uint32))
        ring, drc, filt = np.zeros(1280, np.int32), 0, 0
        for f in range(FRAMES):
            re, im = chain_input(1, c, f)
            o, drc, filt = t.syn(os_, re, im, ring, drc, filt)
            assert (crc(o), crc(ring), (drc << 16) | filt) == tuple(int(x) for x in GOLD["syn_crc"][c, f]), (c, f)
        assert np.array_equal(o.view(np.uint32), GOLD["syn_last"][c].view(np.uint32))


@pytest.mark.parametrize("nb,slots", NB_BANKS)
def test_oracle_matches_reference_chains_of_the_8_3_and_4_1_banks(oracle, nb, slots):
    fn = t._bind_nb(oracle.lib, "xo")
    for c in range(CHAINS):
        ring, pos, win = np.zeros(320, np.int32), 0, 0
        for f in range(FRAMES):
            re, im, pos, win = t.ana_nb(fn, chain_input_nb(nb, c, f), nb, slots, ring, pos, win)
            assert (crc(re), crc(im), crc(ring), (pos << 16) | win) == tuple(int(x) for x in GOLD["ana%d_crc" % nb][c, f]), (c, f)
        assert np.array_equal(re.view(np.uint32), GOLD["ana%d_last" % nb][c, 0].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("nb,slots", NB_BANKS)
def test_gpu_analysis_chains_of_the_8_3_and_4_1_banks(nb, slots):
    """xaac_esbr_qmf_analysis_nb_batch (24 channels x 32 slots, 16 x 64) on the reference-made chains, all chains as one batch
    with the core rows 1024 floats apart: float words, rings and positions identical, bands nb..31 zeroed, 32..63 untouched"""
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = CHAINS + 1
    state = torch.zeros((n, libxaac_amd.ESBR_ANA_STATE_WORDS), dtype=torch.int32, device=dev)
    re = torch.full((n, slots, 64), 7.0, dtype=torch.float32, device=dev)
    im = torch.full((n, slots, 64), 7.0, dtype=torch.float32, device=dev)
    for f in range(FRAMES):
        core = np.full((n, 1024), 3.0, np.float32)
        for c in range(n):
            core[c, :nb * slots] = chain_input_nb(nb, c % CHAINS, f)
        ctx.esbr_qmf_analysis_nb_batch(nb, slots, torch.from_numpy(core).to(dev), state, re, im)
        ctx.sync()
        r, i, st = re.cpu().numpy(), im.cpu().numpy(), state.cpu().numpy()
        assert np.all(r[:, :, 32:] == 7.0) and np.all(i[:, :, 32:] == 7.0)
        assert not np.any(r[:, :, nb:32]) and not np.any(i[:, :, nb:32])
        r[:, :, 32:] = 0
        i[:, :, 32:] = 0
        for c in range(CHAINS):
            got = (crc(r[c]), crc(i[c]), crc(st[c, :320]), (int(st[c, 320]) << 16) | int(st[c, 321]))
            assert got == tuple(int(x) for x in GOLD["ana%d_crc" % nb][c, f]), (c, f)
        assert np.array_equal(r[CHAINS], r[0]) and np.array_equal(st[CHAINS], st[0])
    assert np.array_equal(r[:CHAINS].view(np.uint32), GOLD["ana%d_last" % nb][:, 0].view(np.uint32))


@pytest.mark.gpu
def test_gpu_analysis_chains():
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = CHAINS + 1  # one more channel fed with chain 0's frames shifted by one: an odd-sized batch of mixed positions
    state = torch.zeros((n, libxaac_amd.ESBR_ANA_STATE_WORDS), dtype=torch.int32, device=dev)
    re = torch.full((n, 32, 64), 7.0, dtype=torch.float32, device=dev)
    im = torch.full((n, 32, 64), 7.0, dtype=torch.float32, device=dev)
    for f in range(FRAMES):
        core = np.stack([chain_input(0, c, f) for c in range(CHAINS)] + [chain_input(0, 0, f)])
        ctx.esbr_qmf_analysis_batch(torch.from_numpy(core).to(dev), state, re, im)
        ctx.sync()
        r, i, st = re.cpu().numpy(), im.cpu().numpy(), state.cpu().numpy()
        assert np.all(r[:, :, 32:] == 7.0) and np.all(i[:, :, 32:] == 7.0)  # bands 32..63 are not the bank's to write
        r[:, :, 32:] = 0
        i[:, :, 32:] = 0
        for c in range(CHAINS):
            got = (crc(r[c]), crc(i[c]), crc(st[c, :320]), (int(st[c, 320]) << 16) | int(st[c, 321]))
            assert got == tuple(int(x) for x in GOLD["ana_crc"][c, f]), (c, f)
        assert np.array_equal(r[CHAINS], r[0]) and np.array_equal(st[CHAINS], st[0])
    assert np.array_equal(r[:CHAINS].view(np.uint32), GOLD["ana_last"][:, 0].view(np.uint32))


@pytest.mark.gpu
def test_gpu_synthesis_chains():
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = CHAINS + 1
    state = torch.zeros((n, libxaac_amd.ESBR_SYN_STATE_WORDS), dtype=torch.int32, device=dev)
    out = torch.zeros((n, 2048), dtype=torch.float32, device=dev)
    for f in range(FRAMES):
        ins = [chain_input(1, c, f) for c in range(CHAINS)] + [chain_input(1, 0, f)]
        re = torch.from_numpy(np.stack([x[0] for x in ins])).to(dev)
        im = torch.from_numpy(np.stack([x[1] for x in ins])).to(dev)
        ctx.esbr_qmf_synthesis_batch(re, im, state, out)
        ctx.sync()
        o, st = out.cpu().numpy(), state.cpu().numpy()
        for c in range(CHAINS):
            got = (crc(o[c]), crc(st[c, :1280]), (int(st[c, 1280]) << 16) | int(st[c, 1281]))
            assert got == tuple(int(x) for x in GOLD["syn_crc"][c, f]), (c, f)
        assert np.array_equal(o[CHAINS], o[0])
    assert np.array_equal(o[:CHAINS].view(np.uint32), GOLD["syn_last"].view(np.uint32))


@pytest.mark.gpu
def test_gpu_analysis_then_synthesis_vs_oracle(oracle):
    """a larger batch (777 channels) of random frames, analysis output fed straight into synthesis, three frames with
    the states carried on both sides"""
    import torch
    import libxaac_amd
    oa, os_ = t._bind(oracle.lib, "xo")
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = 777
    rng = np.random.default_rng(11)
    sa = torch.zeros((n, libxaac_amd.ESBR_ANA_STATE_WORDS), dtype=torch.int32, device=dev)
    ss = torch.zeros((n, libxaac_amd.ESBR_SYN_STATE_WORDS), dtype=torch.int32, device=dev)
    re = torch.zeros((n, 32, 64), dtype=torch.float32, device=dev)
    im = torch.zeros((n, 32, 64), dtype=torch.float32, device=dev)
    out = torch.zeros((n, 2048), dtype=torch.float32, device=dev)
    check = [0, 1, 2, 333, 775, 776]
    o_ana = {c: [np.zeros(320, np.int32), 0, 0] for c in check}
    o_syn = {c: [np.zeros(1280, np.int32), 0, 0] for c in check}
    for f in range(3):
        core = (rng.uniform(-1, 1, (n, 1024)) * rng.choice([32768.0, 3000.0, 20.0], (n, 1))).astype(np.float32)
        ctx.esbr_qmf_analysis_batch(torch.from_numpy(core).to(dev), sa, re, im)
        ctx.esbr_qmf_synthesis_batch(re, im, ss, out)
        ctx.sync()
        r, i, o = re.cpu().numpy(), im.cpu().numpy(), out.cpu().numpy()
        a_st, s_st = sa.cpu().numpy(), ss.cpu().numpy()
        for c in check:
            ring, pos, win = o_ana[c]
            xr, xi, pos, win = t.ana(oa, np.ascontiguousarray(core[c]), ring, pos, win)
            o_ana[c][1:] = [pos, win]
            assert np.array_equal(xr.view(np.uint32), r[c].view(np.uint32)), (f, c)
            assert np.array_equal(xi.view(np.uint32), i[c].view(np.uint32)), (f, c)
            assert np.array_equal(a_st[c], np.concatenate([ring, [pos, win]])), (f, c)
            ring, drc, filt = o_syn[c]
            xo, drc, filt = t.syn(os_, xr, xi, ring, drc, filt)
            o_syn[c][1:] = [drc, filt]
            assert np.array_equal(xo.view(np.uint32), o[c].view(np.uint32)), (f, c)
            assert np.array_equal(s_st[c], np.concatenate([ring, [drc, filt]])), (f, c)
        assert np.any(o != 0)


@pytest.mark.gpu
def test_gpu_down_sampled_synthesis_chain_equals_the_oracle(oracle):
    """xaac_esbr_qmf_synthesis_ds_batch (32 synthesis channels: sbr_dec.c:605-628) against the oracle's restatement -- itself pinned on
    the reference's function, tests/test_esbr_qmf_oracle_vs_reference.py -- over 12 frames with the state carried on the device: an odd
    number of channels (the last wave half empty), several signal levels, ring and window positions wrapping"""
    import ctypes
    import torch
    import libxaac_amd
    PF, P32 = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
    fn = oracle.lib.xo_esbr_synthesis_ds
    fn.restype = None
    fn.argtypes = [PF, PF, P32, P32, P32, PF]
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = 7
    rng = np.random.default_rng(21)
    state = torch.zeros((n, libxaac_amd.ESBR_SYN_STATE_WORDS), dtype=torch.int32, device=dev)
    rings = [np.zeros(1280, np.int32) for _ in range(n)]
    pos = [[0, 0] for _ in range(n)]
    out = torch.full((n, 1024), 5.0, dtype=torch.float32, device=dev)
    for f in range(12):
        amp = np.float32([1.0, 30.0, 1e-3, 4000.0, 3e8, 200.0, 0.0][(f + 1) % 7])
        re = (rng.standard_normal((n, 32, 64)) * amp).astype(np.float32)
        im = (rng.standard_normal((n, 32, 64)) * amp).astype(np.float32)
        ctx.esbr_qmf_synthesis_ds_batch(torch.from_numpy(re).to(dev), torch.from_numpy(im).to(dev), state, out)
        ctx.sync()
        o, st = out.cpu().numpy(), state.cpu().numpy()
        for c in range(n):
            want = np.zeros(2048, np.float32)
            d, fl = ctypes.c_int32(pos[c][0]), ctypes.c_int32(pos[c][1])
            fn(re[c].ctypes.data_as(PF), im[c].ctypes.data_as(PF), rings[c].ctypes.data_as(P32), ctypes.byref(d), ctypes.byref(fl),
               want.ctypes.data_as(PF))
            pos[c] = [d.value, fl.value]
            assert np.array_equal(o[c].view(np.uint32), want[:1024].view(np.uint32)), (f, c, int(np.sum(o[c] != want[:1024])))
            assert np.array_equal(st[c, :640], rings[c][:640]) and (int(st[c, 1280]), int(st[c, 1281])) == (d.value, fl.value), (f, c)
