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
