"""xaac_qmf_analysis_eld_batch (the LD / ELD complex analysis bank) against the oracle's literal restatement, which
tests/test_qmf_eld_oracle_vs_reference.py pins on the compiled reference: batches of 16- and 15-slot frames, chains with
the ring and the four pointers carried on the device, channels in different phases of the pointer cycle in one batch, a
channel whose pointers are out of step (refused)."""
import ctypes

import numpy as np
import pytest

P16 = ctypes.POINTER(ctypes.c_int16)
P32 = ctypes.POINTER(ctypes.c_int32)


@pytest.mark.gpu
@pytest.mark.parametrize("n_slots", [16, 15])
def test_eld_analysis_chain_vs_oracle(oracle, n_slots):
    import torch
    import libxaac_amd
    fn = oracle.lib.xo_qmf_analysis_eld
    fn.restype = None
    fn.argtypes = [P16, ctypes.c_int, P16, P16, ctypes.c_int, ctypes.c_int, P32, ctypes.c_int]
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = 11   # not a multiple of four: the last wave is partly empty
    rng = np.random.default_rng(60 + n_slots)
    host = np.zeros((n, 324), np.int16)
    host[:, 322] = 32
    # channels 1..9 start somewhere in the pointer cycle: run the oracle a few frames ahead on them
    for ch in range(1, n - 1):
        for _ in range(ch):
            pcm = rng.integers(-20000, 20000, 32 * n_slots).astype(np.int16)
            q = np.zeros((n_slots, 128), np.int32)
            fn(pcm.ctypes.data_as(P16), 1, host[ch, :320].ctypes.data_as(P16), host[ch, 320:].ctypes.data_as(P16), n_slots, 20,
               q.ctypes.data_as(P32), 128)
    host[n - 1, 323] = 32   # fp out of step with wr
    state = torch.from_numpy(host.copy()).to(dev)
    status = torch.full((n,), 7, dtype=torch.int32, device=dev)
    for frame in range(12):
        amp = [32767, 2500, 9][frame % 3]
        pcm = rng.integers(-amp, amp + 1, (n, 32 * n_slots)).astype(np.int16)
        if frame == 5:
            pcm[:] = -32768
        usb = int(rng.integers(0, 33))
        qmf = torch.full((n, n_slots, 128), 5, dtype=torch.int32, device=dev)
        ctx.qmf_analysis_eld_batch(torch.from_numpy(pcm).to(dev), state, qmf, n_slots, usb, status)
        ctx.sync()
        got, gs = qmf.cpu().numpy(), state.cpu().numpy()
        assert status.cpu().tolist() == [0] * (n - 1) + [-1]
        for ch in range(n - 1):
            q = np.full((n_slots, 128), 5, np.int32)
            fn(np.ascontiguousarray(pcm[ch]).ctypes.data_as(P16), 1, host[ch, :320].ctypes.data_as(P16), host[ch, 320:].ctypes.data_as(P16),
               n_slots, usb, q.ctypes.data_as(P32), 128)
            assert np.array_equal(q, got[ch]), (frame, ch)
            assert np.array_equal(host[ch], gs[ch]), (frame, ch, host[ch, 320:], gs[ch, 320:])
        assert np.array_equal(gs[n - 1], host[n - 1]) and np.all(got[n - 1] == 5)


@pytest.mark.gpu
@pytest.mark.parametrize("n_slots", [16, 15])
def test_eld_synthesis_chain_vs_oracle(oracle, n_slots):
    """xaac_qmf_synthesis_eld_batch against the oracle's literal pointer-rotating form: PCM, ring and the four state words
    over chains of frames carried on the device, channels in different phases of the ten-slot cycle in one batch, region
    scales varied per frame, levels up to clipping, one channel with an impossible state (refused and left alone)"""
    import torch
    import libxaac_amd
    fn = oracle.lib.xo_qmf_synthesis_eld
    fn.restype = None
    fn.argtypes = [P32, ctypes.c_int, P16, ctypes.c_int, ctypes.c_int, ctypes.c_int, P16, P16, ctypes.c_int, P16, ctypes.c_int]
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = 11
    rng = np.random.default_rng(80 + n_slots)
    host = np.zeros((n, 1284), np.int16)
    host[:, 1283] = 64

    def run_oracle(ch, q, sf, lsb, usb, split, pcm):
        qq = np.ascontiguousarray(q)
        fn(qq.ctypes.data_as(P32), 128, sf.ctypes.data_as(P16), lsb, usb, split, host[ch, :1280].ctypes.data_as(P16),
           host[ch, 1280:].ctypes.data_as(P16), n_slots, pcm.ctypes.data_as(P16), 1)

    for ch in range(1, n - 1):   # different phases (15-slot frames walk through all ten; 16-slot ones through five)
        for _ in range(ch):
            q = (rng.standard_normal((n_slots, 128)) * 2.0 ** 22).astype(np.int32)
            run_oracle(ch, q, np.array([-3, -3, -3, -2], np.int16), 20, 40, 3, np.zeros(64 * n_slots, np.int16))
    host[n - 1, 1281] = 64   # phase out of step with drc_offset
    state = torch.from_numpy(host.copy()).to(dev)
    status = torch.full((n,), 7, dtype=torch.int32, device=dev)
    for frame in range(12):
        level = 2.0 ** rng.integers(8, 30)
        q = (rng.standard_normal((n, n_slots, 128)) * level).clip(-2 ** 31, 2 ** 31 - 1).astype(np.int32)
        if frame % 6 == 5:
            q[:] = 2 ** 31 - 1 if frame % 2 else -2 ** 31
        sf = rng.integers(-12, 3, (n, 4)).astype(np.int16)
        lsb = int(rng.integers(0, 40))
        usb = int(rng.integers(lsb, 65))
        split = int(rng.integers(0, n_slots + 1))
        pcm = torch.full((n, 64 * n_slots), 5, dtype=torch.int16, device=dev)
        qd = torch.from_numpy(q).to(dev)
        scaled = torch.full((n, n_slots, 128), 9, dtype=torch.int32, device=dev)
        ctx.qmf_synthesis_eld_batch(qd, torch.from_numpy(sf).to(dev), state, pcm, n_slots, lsb, usb, split, status, scaled)
        ctx.sync()
        rs = oracle.lib.xo_qmf_eld_region_scale
        rs.restype = None
        rs.argtypes = [P32, ctypes.c_int, P16, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P32]
        gsc = scaled.cpu().numpy()
        for ch in range(n - 1):
            want_sc = np.zeros((n_slots, 128), np.int32)
            rs(np.ascontiguousarray(q[ch]).ctypes.data_as(P32), 128, np.ascontiguousarray(sf[ch]).ctypes.data_as(P16), lsb, usb, split, n_slots,
               want_sc.ctypes.data_as(P32))
            assert np.array_equal(gsc[ch], want_sc), (frame, ch)
        assert np.all(gsc[n - 1] == 9)
        assert np.array_equal(qd.cpu().numpy(), q)
        got, gs = pcm.cpu().numpy(), state.cpu().numpy()
        assert status.cpu().tolist() == [0] * (n - 1) + [-1]
        for ch in range(n - 1):
            want = np.zeros(64 * n_slots, np.int16)
            run_oracle(ch, q[ch], sf[ch], lsb, usb, split, want)
            assert np.array_equal(want, got[ch]), (frame, ch, np.nonzero(want != got[ch])[0][:5])
            assert np.array_equal(host[ch], gs[ch]), (frame, ch, host[ch, 1280:], gs[ch, 1280:])
        assert np.array_equal(gs[n - 1], host[n - 1]) and np.all(got[n - 1] == 5)
