"""tests/golden/qmf_eld_ref.npz: chains of the REAL LD / ELD complex QMF banks (ixheaacd_cplx_anal_qmffilt /
ixheaacd_cplx_synt_qmffilt with AOT_ER_AAC_ELD) made by tools/make_golden_qmf_eld.py: 10 chains x 26 frames of 16 and of
15 slots per bank, ring and pointer state carried, inputs regenerated from integer counters.
  * CPU: the oracle's literal restatement walks the chains and reproduces every CRC (output, ring, state words);
  * GPU (-m gpu): xaac_qmf_analysis_eld_batch / xaac_qmf_synthesis_eld_batch walk all chains of a frame length as one batch
    with the state on the device -- the time-invariant kernel forms meet reference data directly."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden_qmf_eld import CHAINS, FRAMES, crc, eld_pcm, eld_qmf, params  # noqa: E402  (input generators: data, not reference code)

P16 = ctypes.POINTER(ctypes.c_int16)
P32 = ctypes.POINTER(ctypes.c_int32)
G = np.load(os.path.join(ROOT, "tests", "golden", "qmf_eld_ref.npz"))


@pytest.mark.parametrize("n_slots", [16, 15])
def test_oracle_walks_the_reference_chains(oracle, n_slots):
    ana, syn = oracle.lib.xo_qmf_analysis_eld, oracle.lib.xo_qmf_synthesis_eld
    ana.restype = syn.restype = None
    ana.argtypes = [P16, ctypes.c_int, P16, P16, ctypes.c_int, ctypes.c_int, P32, ctypes.c_int]
    syn.argtypes = [P32, ctypes.c_int, P16, ctypes.c_int, ctypes.c_int, ctypes.c_int, P16, P16, ctypes.c_int, P16, ctypes.c_int]
    for c in range(CHAINS):
        ring_a, st_a = np.zeros(320, np.int16), np.array([0, 0, 32, 0], np.int16)
        ring_s, st_s = np.zeros(1280, np.int16), np.array([0, 0, 0, 64], np.int16)
        for f in range(FRAMES):
            usb_a, sf, lsb, usb, split = params(n_slots, c, f)
            pcm = eld_pcm(n_slots, c, f)
            q = np.full((n_slots, 128), 5, np.int32)
            ana(pcm.ctypes.data_as(P16), 1, ring_a.ctypes.data_as(P16), st_a.ctypes.data_as(P16), n_slots, usb_a, q.ctypes.data_as(P32), 128)
            assert (crc(q), crc(ring_a), crc(st_a)) == tuple(G["ana_crc_%d" % n_slots][c, f]), ("analysis", c, f)
            qq = eld_qmf(n_slots, c, f).copy()
            out = np.zeros(64 * n_slots, np.int16)
            syn(qq.ctypes.data_as(P32), 128, sf.ctypes.data_as(P16), lsb, usb, split, ring_s.ctypes.data_as(P16), st_s.ctypes.data_as(P16),
                n_slots, out.ctypes.data_as(P16), 1)
            assert (crc(out), crc(ring_s), crc(st_s)) == tuple(G["syn_crc_%d" % n_slots][c, f]), ("synthesis", c, f)


@pytest.mark.gpu
@pytest.mark.parametrize("n_slots", [16, 15])
def test_gpu_walks_the_reference_chains(n_slots):
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = CHAINS
    host_a = np.zeros((n, 324), np.int16)
    host_a[:, 322] = 32
    host_s = np.zeros((n, 1284), np.int16)
    host_s[:, 1283] = 64
    st_a, st_s = torch.from_numpy(host_a).to(dev), torch.from_numpy(host_s).to(dev)
    status = torch.full((n,), 7, dtype=torch.int32, device=dev)
    for f in range(FRAMES):
        # the batch entries take one usb / lsb / usb / split per launch: the chains' own values differ, so one launch per chain
        # value set would defeat the batch -- the kernels take them per launch, hence a launch per chain on a 1-row view
        for c in range(n):
            usb_a, sf, lsb, usb, split = params(n_slots, c, f)
            pcm = torch.from_numpy(eld_pcm(n_slots, c, f)[None]).to(dev)
            qmf = torch.full((1, n_slots, 128), 5, dtype=torch.int32, device=dev)
            sa = st_a[c:c + 1]
            ctx.qmf_analysis_eld_batch(pcm, sa, qmf, n_slots, usb_a, status[c:c + 1])
            q = torch.from_numpy(eld_qmf(n_slots, c, f)[None].copy()).to(dev)
            out = torch.full((1, 64 * n_slots), 5, dtype=torch.int16, device=dev)
            ss = st_s[c:c + 1]
            ctx.qmf_synthesis_eld_batch(q, torch.from_numpy(sf[None]).to(dev), ss, out, n_slots, lsb, usb, split, status[c:c + 1])
            ctx.sync()
            a, s = sa.cpu().numpy()[0], ss.cpu().numpy()[0]
            assert status[c].item() == 0
            assert (crc(qmf.cpu().numpy()[0]), crc(a[:320]), crc(a[320:])) == tuple(G["ana_crc_%d" % n_slots][c, f]), ("analysis", c, f)
            assert (crc(out.cpu().numpy()[0]), crc(s[:1280]), crc(s[1280:])) == tuple(G["syn_crc_%d" % n_slots][c, f]), ("synthesis", c, f)
    ctx.close()
