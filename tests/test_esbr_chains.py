"""tests/golden/esbr_chains.npz: 1451 calls of the REAL ixheaacd_sbr_dec taking its eSBR branch ("Path A": the reference's
default -esbr:1 path for HE-AAC / HE-AACv2), made by tools/make_golden_esbr_chains.py as 48 chains (one channel of one run
of the reference decoder over a committed stream) with reference-side fuzz of the live side info -- limiter gains and
bands, interpolation, smoothing, inverse-filter modes, added harmonics, inter-TES, harmonic patching with and without a
pitch, reset frames, PS quantiser / 1-4 envelopes / IID / ICC -- and the state carried by the reference itself.  Stored
per step: side info in the boundary formats, the return code, CRC32s of out / out_r and of the eSBR, transposer and PS
states after the call; the float core input is regenerated here (counter-based integer generator).
  * CPU: the oracle's whole-frame function walks every chain (its own state carried) and must reproduce every CRC;
  * GPU (-m gpu): xaac_esbr_sbr_process_batch walks all chains as one batch with the three states resident on the device:
    the device builds of the float HF generator, envelope adjuster, float PS and transposer-in-chain meet reference data
    directly (the float words, not +-1 LSB), not via the oracle."""
import ctypes
import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden_esbr_chains import chain_core  # noqa: E402  (the generator's own input function: data, not reference code)

PF = ctypes.POINTER(ctypes.c_float)
FIXTURES = {"aac": np.load(os.path.join(ROOT, "tests", "golden", "esbr_chains.npz")),
            # the same chains for USAC channels (tools/make_golden_esbr_chains.py usac: stereo 2:1 eSBR with codec_x_delay 0, with the
            # harmonic transposer, a switched FD / LPD core's and a PVC stream's ORIG_SBR frames; 1482 calls in 78 chains)
            "usac": np.load(os.path.join(ROOT, "tests", "golden", "esbr_usac_chains.npz"))}
CH = FIXTURES["aac"]
NO_X_DELAY, USAC = 8, 4     # include/xaac_esbr.h: bits of xaac_esbr_side::harmonic_sbr


def chain_has_transposer(CH, rows):
    """a USAC channel without a harmonic transposer (XAAC_ESBR_NO_X_DELAY in its side info) is run without an hbe_state"""
    from esbr_structs import EsbrSide
    return not (int(CH["side"][rows[0]].view(np.int16)[EsbrSide.harmonic_sbr.offset // 2]) & NO_X_DELAY)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def state_crc(st, header, frame, apply):
    """CRC32 of an xaac_esbr_state without the sbr_qmf_out history entries no later call can read (rows from the frame's
    last border on, bands below the cross-over): the reference's persistent 64-row matrix holds stale values there, a
    frame-by-frame implementation zeros -- oracle/ref_capture.c masks the same entries before its CRC"""
    from esbr_structs import EsbrState
    import sbr_capture as cap
    h = cap.Header.from_buffer_copy(header.tobytes())
    f = cap.Frame.from_buffer_copy(frame.tobytes())
    keep = 2 + 2 * f.border_vec[f.num_env] - 32 if apply else 8
    b = np.array(st, copy=True)
    for name in ("out_re", "out_im"):
        fld = getattr(EsbrState, name)
        m = b[fld.offset:fld.offset + fld.size].view(np.float32).reshape(8, 64)
        m[max(keep, 0):, :] = 0
        m[:, :h.sub_band_start] = 0
    return crc(b)


def steps_of_chains(CH=CH):
    sc, si = CH["step_chain"], CH["step_idx"]
    order = [np.nonzero(sc == c)[0] for c in range(len(CH["chain_len"]))]
    for c, rows in enumerate(order):
        assert np.array_equal(si[rows], np.arange(len(rows))) and len(rows) == CH["chain_len"][c]
    return order


def test_fixture_is_what_it_says():
    assert CH["ret"].size >= 1000 and not CH["ret"].any()
    from esbr_structs import EsbrSide
    harm = CH["side"].view(np.int16)[:, EsbrSide.harmonic_sbr.offset // 2]
    pitch = CH["side"].view(np.int32)[:, EsbrSide.pitch_in_bins.offset // 4]
    proc = CH["apply"] != 0
    assert ((harm[proc] & 1) != 0).sum() > 300 and (((harm & 1) != 0) & (pitch != 0) & proc).sum() > 100
    assert (harm[proc] == 2).sum() > 300     # XAAC_ESBR_PRE_FLATTEN on frames with LPP patches
    assert (CH["chain_ps"] != 0).sum() >= 4


def test_usac_fixture_is_what_it_says():
    U = FIXTURES["usac"]
    from esbr_structs import EsbrSide
    flags = U["side"].view(np.int16)[:, EsbrSide.harmonic_sbr.offset // 2]
    assert U["ret"].size >= 1000 and not U["ret"].any() and (flags & USAC).all()
    assert ((flags & NO_X_DELAY) != 0).sum() > 500 and ((flags & NO_X_DELAY) == 0).sum() > 200      # without / with a transposer
    assert ((flags & 1) != 0).sum() > 100                                                            # harmonic patching frames
    from esbr_structs import EsbrPvcSide
    mode = U["pvc_side"].view(np.int16)[:, EsbrPvcSide.sbr_mode.offset // 2]
    assert (mode == 2).sum() > 150 and (mode == 1).sum() > 1000 and (mode == 0).sum() > 0      # PVC_SBR, ORIG_SBR, UNKNOWN_SBR frames


@pytest.mark.parametrize("which", ["aac", "usac"])
def test_oracle_walks_the_reference_chains(oracle, which):
    CH = FIXTURES[which]
    fn = oracle.lib.xo_esbr_sbr_frame_pvc
    fn.restype = ctypes.c_int
    fn.argtypes = [PF] + [ctypes.c_void_p] * 6 + [PF, PF, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    pvc = "pvc_side" in CH.files     # the USAC fixture: PVC side info per step, PVC state per chain, a sixth CRC
    for c, rows in enumerate(steps_of_chains(CH)):
        run, cid, eps = int(CH["chain_run"][c]), int(CH["chain_id"][c]), bool(CH["chain_ps"][c])
        st, hb, ps = CH["est0"][c].copy(), CH["hbs0"][c].copy(), CH["eps0"][c].copy()
        with_hb = chain_has_transposer(CH, rows)
        pv = CH["pvst0"][c].copy() if pvc else None
        for s, r in enumerate(rows):
            core = np.ascontiguousarray(chain_core(run, cid, s))
            out, out_r = np.zeros(2048, np.float32), np.zeros(2048, np.float32)
            h, f, sd, pf = (np.ascontiguousarray(CH[k][r]) for k in ("header", "frame", "side", "ps_frame"))
            rc = fn(core.ctypes.data_as(PF), vp(h), vp(f), vp(sd), vp(st), vp(pf) if eps else None, vp(ps) if eps else None,
                    out.ctypes.data_as(PF), out_r.ctypes.data_as(PF) if eps else None, vp(hb) if with_hb else None,
                    vp(np.ascontiguousarray(CH["pvc_side"][r])) if pvc else None, vp(pv) if pvc else None)
            want = CH["crc"][r]
            assert rc == CH["ret"][r], (c, s)
            assert crc(out) == want[0], ("out", c, s)
            if eps and CH["apply"][r]:
                assert crc(out_r) == want[1], ("out_r", c, s)
            assert state_crc(st, h, f, CH["apply"][r]) == want[2], ("state", c, s)
            assert crc(hb) == want[3], ("transposer state", c, s)
            if eps:
                assert crc(ps) == want[4], ("ps state", c, s)
            if pvc:
                assert crc(pv) == want[5], ("pvc state", c, s)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["aac", "usac"])
def test_gpu_walks_the_reference_chains(which):
    import torch
    import libxaac_amd
    CH = FIXTURES[which]
    ctx = libxaac_amd.XaacContext(0, 0)
    dev = torch.device("cuda:0")
    order = steps_of_chains(CH)
    # the PS streams form their own batch (ps_frame / ps_state / out_r are per launch), and so do the channels without a
    # harmonic transposer (hbe_state is per launch)
    for with_ps, with_hb in ((False, True), (True, True), (False, False)):
        chains = [c for c in range(len(order)) if bool(CH["chain_ps"][c]) == with_ps and chain_has_transposer(CH, order[c]) == with_hb]
        n = len(chains)
        if n == 0:
            continue
        t_st = torch.from_numpy(np.ascontiguousarray(CH["est0"][chains])).to(dev)
        t_hb = torch.from_numpy(np.ascontiguousarray(CH["hbs0"][chains])).to(dev)
        t_ps = torch.from_numpy(np.ascontiguousarray(CH["eps0"][chains])).to(dev) if with_ps else None
        pvc = "pvc_side" in CH.files
        t_pv = torch.from_numpy(np.ascontiguousarray(CH["pvst0"][chains])).to(dev) if pvc else None
        for s in range(max(len(order[c]) for c in chains)):
            act = [i for i, c in enumerate(chains) if s < len(order[c])]      # chains that still have a step s
            rows = [order[chains[i]][s] for i in act]
            m = len(act)
            idx = torch.tensor(act, device=dev)
            core = torch.from_numpy(np.stack([chain_core(int(CH["chain_run"][chains[i]]), int(CH["chain_id"][chains[i]]), s) for i in act])).to(dev)
            g = lambda k: torch.from_numpy(np.ascontiguousarray(CH[k][rows])).to(dev)
            st, hb = t_st[idx].contiguous(), t_hb[idx].contiguous()
            ps = t_ps[idx].contiguous() if with_ps else None
            pv = t_pv[idx].contiguous() if pvc else None
            out = torch.zeros((m, 2048), dtype=torch.float32, device=dev)
            out_r = torch.zeros((m, 2048), dtype=torch.float32, device=dev) if with_ps else None
            status = torch.full((m,), 7, dtype=torch.int32, device=dev)
            ws = torch.zeros(ctx.esbr_workspace_bytes(m), dtype=torch.uint8, device=dev)
            ctx.esbr_sbr_process_batch(core, g("header"), g("frame"), g("side"), st, out, ws, status,
                                       ps_frame=g("ps_frame") if with_ps else None, ps_state=ps, out_r=out_r,
                                       hbe_state=hb if with_hb else None, pvc_side=g("pvc_side") if pvc else None, pvc_state=pv)
            ctx.sync()
            t_st[idx], t_hb[idx] = st, hb
            if pvc:
                t_pv[idx] = pv
            if with_ps:
                t_ps[idx] = ps
            assert np.array_equal(status.cpu().numpy(), CH["ret"][rows]), s
            o, orr = out.cpu().numpy(), (out_r.cpu().numpy() if with_ps else None)
            stn, hbn, psn = st.cpu().numpy(), hb.cpu().numpy(), (ps.cpu().numpy() if with_ps else None)
            pvn = pv.cpu().numpy() if pvc else None
            for j, r in enumerate(rows):
                want = CH["crc"][r]
                assert crc(o[j]) == want[0], ("out", chains[act[j]], s)
                if with_ps and CH["apply"][r]:
                    assert crc(orr[j]) == want[1], ("out_r", chains[act[j]], s)
                assert state_crc(stn[j], CH["header"][r], CH["frame"][r], CH["apply"][r]) == want[2], ("state", chains[act[j]], s)
                assert crc(hbn[j]) == want[3], ("transposer state", chains[act[j]], s)
                if with_ps:
                    assert crc(psn[j]) == want[4], ("ps state", chains[act[j]], s)
                if pvc:
                    assert crc(pvn[j]) == want[5], ("pvc state", chains[act[j]], s)
    ctx.close()


@pytest.mark.gpu
def test_gpu_on_fuzzed_grids_and_a_moving_band_limit_equals_the_oracle(oracle):
    """the first six frames of every chain with every second frame's envelope grid fuzzed (parser-like, variable, anything in
    0..19: unsorted, empty, behind slot 32, ending before slot 16) and the band limit moving -- side info no parser makes,
    which the boundary has to contain: the GPU chain (three states on the device) against the oracle frame by frame: return
    codes, output words, the states' checksums.  (tests/test_sbr_core_sanitized.py runs the same frames under ASan.)"""
    import torch
    import libxaac_amd
    import sbr_capture as cap
    from test_env_pairs_cpu import _fuzz_frame
    fn = oracle.lib.xo_esbr_sbr_frame_hbe
    fn.restype = ctypes.c_int
    fn.argtypes = [PF] + [ctypes.c_void_p] * 6 + [PF, PF, ctypes.c_void_p]
    ctx = libxaac_amd.XaacContext(0, 0)
    dev = torch.device("cuda:0")
    order = steps_of_chains()
    rng = np.random.default_rng(int(os.environ.get("XAAC_FUZZ_SEED", "3")))   # (a soak: the same test under other seeds)
    taken = refused = 0
    for with_ps in (False, True):
        chains = [c for c in range(len(order)) if bool(CH["chain_ps"][c]) == with_ps and len(order[c]) >= 6]
        n = len(chains)
        st = [CH["est0"][c].copy() for c in chains]
        hb = [CH["hbs0"][c].copy() for c in chains]
        ps = [CH["eps0"][c].copy() for c in chains]
        for s in range(6):
            hdr, frm, side, psf, cores = [], [], [], [], []
            for i, c in enumerate(chains):
                r = order[c][s]
                h, f = np.ascontiguousarray(CH["header"][r]).copy(), np.ascontiguousarray(CH["frame"][r]).copy()
                hh, ff = cap.Header.from_buffer(h), cap.Frame.from_buffer(f)
                if s % 2 == 1:
                    _fuzz_frame(rng, hh, ff, (c + s) % 3)
                if s % 4 == 3:
                    ff.max_qmf_subband_aac = int(np.clip(ff.max_qmf_subband_aac + rng.integers(-6, 7), hh.sub_band_start, 32))
                hdr.append(h); frm.append(f)
                side.append(np.ascontiguousarray(CH["side"][r])); psf.append(np.ascontiguousarray(CH["ps_frame"][r]))
                cores.append(np.ascontiguousarray(chain_core(int(CH["chain_run"][c]), int(CH["chain_id"][c]), s)))
            up = lambda rows: torch.from_numpy(np.stack(rows)).to(dev)
            t_st, t_hb, t_ps = up(st), up(hb), (up(ps) if with_ps else None)
            out = torch.zeros((n, 2048), dtype=torch.float32, device=dev)
            out_r = torch.zeros((n, 2048), dtype=torch.float32, device=dev) if with_ps else None
            status = torch.full((n,), 7, dtype=torch.int32, device=dev)
            ws = torch.zeros(ctx.esbr_workspace_bytes(n), dtype=torch.uint8, device=dev)
            ctx.esbr_sbr_process_batch(up(cores), up(hdr), up(frm), up(side), t_st, out, ws, status,
                                       ps_frame=up(psf) if with_ps else None, ps_state=t_ps, out_r=out_r, hbe_state=t_hb)
            ctx.sync()
            g_rc, g_out = status.cpu().numpy(), out.cpu().numpy()
            g_outr = out_r.cpu().numpy() if with_ps else None
            g_st, g_hb, g_ps = t_st.cpu().numpy(), t_hb.cpu().numpy(), (t_ps.cpu().numpy() if with_ps else None)
            for i, c in enumerate(chains):
                o, orr = np.zeros(2048, np.float32), np.zeros(2048, np.float32)
                rc = fn(cores[i].ctypes.data_as(PF), vp(hdr[i]), vp(frm[i]), vp(side[i]), vp(st[i]), vp(psf[i]) if with_ps else None,
                        vp(ps[i]) if with_ps else None, o.ctypes.data_as(PF), orr.ctypes.data_as(PF) if with_ps else None, vp(hb[i]))
                assert g_rc[i] == rc, ("rc", c, s, int(g_rc[i]), rc)
                if rc != 0:        # a refused frame: the chain goes on from the device's states
                    st[i], hb[i] = g_st[i].copy(), g_hb[i].copy()
                    if with_ps:
                        ps[i] = g_ps[i].copy()
                    refused += 1
                    continue
                taken += 1
                apply = cap.Frame.from_buffer_copy(frm[i].tobytes()).apply_processing
                assert np.array_equal(g_out[i].view(np.uint32), o.view(np.uint32)), ("out", c, s)
                if with_ps and apply:
                    assert np.array_equal(g_outr[i].view(np.uint32), orr.view(np.uint32)), ("out_r", c, s)
                assert state_crc(g_st[i], hdr[i], frm[i], apply) == state_crc(st[i], hdr[i], frm[i], apply), ("state", c, s)
                assert crc(g_hb[i]) == crc(hb[i]), ("transposer state", c, s)
                if with_ps:
                    assert crc(g_ps[i]) == crc(ps[i]), ("ps state", c, s)
    ctx.close()
    assert taken > 150 and refused < taken
