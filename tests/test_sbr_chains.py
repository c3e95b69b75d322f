"""tests/golden/sbr_chains.npz: 1056 + 1056 calls of the REAL ixheaacd_sbr_dec (low-power; HQ + parametric stereo) made
by tools/make_golden_sbr_chains.py as 24 + 24 chains of 44 steps with reference-side fuzz and the state carried by the
reference itself.  Stored per step: side info, the reference's return code, CRC32s of its PCM and of its state(s) after
the call; the core PCM is regenerated here (counter-based integer generator).
  * CPU: the oracle walks every chain (its own state carried) and must reproduce every CRC;
  * GPU (-m gpu): the kernels walk all chains of a kind as one batch, state resident on the device, same CRCs --
    the device-only code paths (DPP scans, ballot closed forms, the frame-at-once PS arrangement, the paired synthesis
    kernel) meet reference data directly, not via the oracle."""
import ctypes
import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden_sbr_chains import chain_pcm  # noqa: E402  (the generator's own PCM function: data, not reference code)

P16 = ctypes.POINTER(ctypes.c_int16)
CH = np.load(os.path.join(ROOT, "tests", "golden", "sbr_chains.npz"))


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("kind", ["lp", "hq"])
def test_oracle_walks_the_reference_chains(oracle, kind):
    hq = kind == "hq"
    hdr, frm, ret = CH[kind + "_header"], CH[kind + "_frame"], CH[kind + "_ret"]
    n_chains, steps = ret.shape
    assert n_chains * steps >= 1000
    for c in range(n_chains):
        st = np.ascontiguousarray(CH[kind + "_st0"][c]).copy()
        ps = np.ascontiguousarray(CH["hq_ps0"][c]).copy() if hq else None
        for s in range(steps):
            pin = np.ascontiguousarray(chain_pcm(1 if hq else 0, c, s))
            h, f = np.ascontiguousarray(hdr[c, s]), np.ascontiguousarray(frm[c, s])
            if hq:
                out = np.zeros(4096, np.int16)
                pf = np.ascontiguousarray(CH["hq_ps_frame"][c, s])
                rc = oracle.lib.xo_sbr_dec_hq(vp(h), vp(f), vp(st), vp(pf), vp(ps), pin.ctypes.data_as(P16), 1,
                                              out.ctypes.data_as(P16), 2)
                assert crc(ps) == CH["hq_crc_ps"][c, s], ("ps state", c, s)
            else:
                out = np.zeros(2048, np.int16)
                rc = oracle.lib.xo_sbr_dec_lp(vp(h), vp(f), vp(st), pin.ctypes.data_as(P16), 1, out.ctypes.data_as(P16), 1)
            assert rc == ret[c, s], (c, s)
            assert crc(out) == CH[kind + "_crc_pcm"][c, s], ("pcm", c, s)
            assert crc(st) == CH[kind + "_crc_state"][c, s], ("state", c, s)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["lp", "hq"])
def test_gpu_walks_the_reference_chains(kind):
    import torch
    import libxaac_amd
    hq = kind == "hq"
    ctx = libxaac_amd.XaacContext(0, 0)
    hdr, frm, ret = CH[kind + "_header"], CH[kind + "_frame"], CH[kind + "_ret"]
    n, steps = ret.shape
    t_st = torch.from_numpy(np.ascontiguousarray(CH[kind + "_st0"])).cuda()
    t_ps = torch.from_numpy(np.ascontiguousarray(CH["hq_ps0"])).cuda() if hq else None
    ws = torch.zeros(ctx.sbr_hq_workspace_bytes(n, True) if hq else ctx.sbr_lp_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    for s in range(steps):
        pin = torch.from_numpy(np.concatenate([chain_pcm(1 if hq else 0, c, s) for c in range(n)])).cuda()
        status = torch.full((n,), 7, dtype=torch.int32, device="cuda")
        th = torch.from_numpy(np.ascontiguousarray(hdr[:, s])).cuda()
        tf = torch.from_numpy(np.ascontiguousarray(frm[:, s])).cuda()
        if hq:
            out = torch.zeros(n * 4096, dtype=torch.int16, device="cuda")
            tpf = torch.from_numpy(np.ascontiguousarray(CH["hq_ps_frame"][:, s])).cuda()
            ctx.sbr_hq_process_batch(pin, th, tf, t_st, out, ws, tpf, t_ps, status)
        else:
            out = torch.zeros(n * 2048, dtype=torch.int16, device="cuda")
            ctx.sbr_lp_process_batch(pin, th, tf, t_st, out, ws, status)
        torch.cuda.synchronize()
        assert np.array_equal(status.cpu().numpy(), ret[:, s]), s
        o = out.cpu().numpy().reshape(n, -1)
        stn = t_st.cpu().numpy()
        for c in range(n):
            assert crc(o[c]) == CH[kind + "_crc_pcm"][c, s], ("pcm", c, s)
            assert crc(stn[c]) == CH[kind + "_crc_state"][c, s], ("state", c, s)
        if hq:
            psn = t_ps.cpu().numpy()
            for c in range(n):
                assert crc(psn[c]) == CH["hq_crc_ps"][c, s], ("ps state", c, s)
    ctx.close()
