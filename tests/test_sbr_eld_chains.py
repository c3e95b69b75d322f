"""tests/golden/sbr_eld_chains.npz: 1256 calls of the REAL ixheaacd_sbr_dec run as an AAC-ELD channel's -- low-delay SBR
(sbr_dec.c:706-775, :1025-1308 with AOT_ER_AAC_ELD): the LD complex analysis bank, block floating point, HF generator and
envelope adjuster on a frame of 16 or 15 QMF slots without overlap slots, the LD complex synthesis bank -- made by
tools/make_golden_sbr_eld_chains.py as 16 chains over the side info of the committed AAC-ELD streams (512- and 480-sample
frames), three of four passes with reference-side fuzz, the state carried by the reference itself.  Per step: side info, the
return code, CRC32s of the PCM, of the state after the call and of the rows the synthesis bank hands on.
  * CPU: the oracle (xo_sbr_dec_eld: libxaac_amd/csrc/sbr_core.h with its low-delay grid) walks every chain;
  * GPU (-m gpu): xaac_sbr_eld_process_batch walks all chains of one frame length as one batch, states resident on the device."""
import ctypes
import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden_sbr_chains import chain_pcm  # noqa: E402  (the generator's own input function: data, not reference code)

CH = np.load(os.path.join(ROOT, "tests", "golden", "sbr_eld_chains.npz"))
P16, P32 = ctypes.POINTER(ctypes.c_int16), ctypes.POINTER(ctypes.c_int32)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def chains():
    return [np.nonzero(CH["step_chain"] == c)[0] for c in range(len(CH["n_slots"]))]


def test_fixture_is_what_it_says():
    assert CH["ret"].size >= 1000 and not CH["ret"].any() and set(CH["n_slots"].tolist()) == {15, 16}
    import sbr_capture as cap
    assert CH["st0"].shape[1] == ctypes.sizeof(cap.EldState)


def test_oracle_walks_the_reference_chains(oracle):
    fn = oracle.lib.xo_sbr_dec_eld
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 3 + [P16, ctypes.c_int, P16, ctypes.c_int, P32]
    for c, rows in enumerate(chains()):
        n = int(CH["n_slots"][c])
        st = CH["st0"][c].copy()
        for s, r in enumerate(rows):
            pin = np.ascontiguousarray(chain_pcm(3, c, s)[:32 * n])
            po, hand = np.zeros(64 * n, np.int16), np.zeros(16 * 128, np.int32)
            h, f = np.ascontiguousarray(CH["header"][r]), np.ascontiguousarray(CH["frame"][r])
            rc = fn(vp(h), vp(f), vp(st), pin.ctypes.data_as(P16), 1, po.ctypes.data_as(P16), 1, hand.ctypes.data_as(P32))
            want = CH["crc"][r]
            assert rc == CH["ret"][r], (c, s)
            assert crc(po) == want[0], ("pcm", c, s)
            assert crc(st) == want[1], ("state", c, s)
            assert crc(hand[:128 * n]) == want[2], ("handed-on rows", c, s)


@pytest.mark.gpu
@pytest.mark.parametrize("n_slots", [16, 15])
def test_gpu_walks_the_reference_chains(n_slots):
    import torch
    import libxaac_amd
    ctx = libxaac_amd.XaacContext(0, 0)
    dev = torch.device("cuda:0")
    order = chains()
    mine = [c for c in range(len(order)) if int(CH["n_slots"][c]) == n_slots]
    assert len(mine) >= 4
    m = len(mine)
    assert CH["st0"].shape[1] == libxaac_amd.SBR_ELD_STATE_BYTES
    t_st = torch.from_numpy(np.ascontiguousarray(CH["st0"][mine])).to(dev)
    ws = torch.zeros(ctx.sbr_eld_workspace_bytes(m), dtype=torch.uint8, device=dev)
    for s in range(min(len(order[c]) for c in mine)):
        rows = [order[c][s] for c in mine]
        pin = torch.from_numpy(np.concatenate([chain_pcm(3, c, s)[:32 * n_slots] for c in mine])).to(dev)
        g = lambda k: torch.from_numpy(np.ascontiguousarray(CH[k][rows])).to(dev)
        out = torch.zeros(m * 64 * n_slots, dtype=torch.int16, device=dev)
        hand = torch.zeros(m * n_slots * 128, dtype=torch.int32, device=dev)
        status = torch.full((m,), 7, dtype=torch.int32, device=dev)
        ctx.sbr_eld_process_batch(pin, g("header"), g("frame"), t_st, out, ws, n_slots, status=status, qmf_handed_on=hand)
        ctx.sync()
        assert np.array_equal(status.cpu().numpy(), CH["ret"][rows]), s
        o, hd, stn = out.cpu().numpy().reshape(m, -1), hand.cpu().numpy().reshape(m, -1), t_st.cpu().numpy()
        for j, r in enumerate(rows):
            want = CH["crc"][r]
            assert crc(o[j]) == want[0], ("pcm", mine[j], s)
            assert crc(stn[j]) == want[1], ("state", mine[j], s)
            assert crc(hd[j]) == want[2], ("handed-on rows", mine[j], s)
    ctx.close()
