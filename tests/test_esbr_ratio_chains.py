"""tests/golden/esbr_ratio_chains.npz: 1140 calls of the REAL ixheaacd_sbr_dec on USAC channels at the two SBR ratios beside 2:1 --
8:3 (768-sample core frames through the 24-channel analysis bank, sbr_dec.c:218) and 4:1 (the 16-channel bank, 64 QMF slots of
which four make an envelope time slot: is_usf_4 in ixheaacd_generate_hf and ixheaacd_sbr_env_calc, pvc_rate 4 in the PVC
decoder) -- made by `tools/make_golden_esbr_chains.py ratios` as 90 chains over streams of the reference encoder (-ccfl_idx:2 / 4,
plain stereo and mono with PVC frames) with the reference-side fuzz of the 2:1 chains and the state carried by the reference.
  * CPU: the oracle's xo_esbr_sbr_frame_ratio walks every chain and reproduces every CRC (output, state, PVC state);
  * GPU (-m gpu): xaac_esbr_sbr_process_batch with sbr_ratio 8:3 / 4:1 walks the chains of a ratio as one batch."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from make_golden_esbr_chains import chain_core  # noqa: E402
from test_esbr_chains import PF, crc, vp, steps_of_chains  # noqa: E402

CH = np.load(os.path.join(ROOT, "tests", "golden", "esbr_ratio_chains.npz"))
RATIO_8_3, RATIO_4_1 = 1, 2          # include/xaac_esbr.h: XAAC_ESBR_RATIO_*
NO_X_DELAY, USAC = 8, 4


def state_crc(st, header, frame, apply, ratio):
    """test_esbr_chains.state_crc with the ratio's slot count: the sbr_qmf_out history rows no later call can read are left out"""
    from esbr_structs import EsbrState
    import sbr_capture as cap
    h = cap.Header.from_buffer_copy(header.tobytes())
    f = cap.Frame.from_buffer_copy(frame.tobytes())
    slots = 64 if ratio == RATIO_4_1 else 32
    hist = 14 if ratio == RATIO_4_1 else 8      # 4:1: rows 8..13 of sbr_qmf_out's history are the state's ph rows 0..5 (xaac_esbr.h)
    keep = 2 + (slots // 16) * f.border_vec[f.num_env] - slots if apply else hist
    b = np.array(st, copy=True)
    for name, first, rows in (("out_re", 0, 8), ("out_im", 0, 8)) + ((("ph_re", 8, 6), ("ph_im", 8, 6)) if hist == 14 else ()):
        fld = getattr(EsbrState, name)
        m = b[fld.offset:fld.offset + fld.size].view(np.float32).reshape(8, 64)[:rows]
        m[max(keep - first, 0):, :] = 0
        m[:, :h.sub_band_start] = 0
    return crc(b)


def test_fixture_is_what_it_says():
    from esbr_structs import EsbrSide, EsbrPvcSide
    assert CH["ret"].size >= 1000 and not CH["ret"].any()
    flags = CH["side"].view(np.int16)[:, EsbrSide.harmonic_sbr.offset // 2]
    assert (flags & USAC).all() and (flags & NO_X_DELAY).all()          # USAC channels without a transposer
    ratio = CH["chain_ratio"][CH["step_chain"]]
    mode = CH["pvc_side"].view(np.int16)[:, EsbrPvcSide.sbr_mode.offset // 2]
    for r in (RATIO_8_3, RATIO_4_1):
        assert (ratio == r).sum() > 300
        assert ((ratio == r) & (mode == 2)).sum() > 50 and ((ratio == r) & (mode == 1)).sum() > 150      # PVC_SBR and ORIG_SBR frames
    assert set(np.unique(CH["chain_ratio"])) == {RATIO_8_3, RATIO_4_1}


def test_oracle_walks_the_reference_chains(oracle):
    fn = oracle.lib.xo_esbr_sbr_frame_ratio
    fn.restype = ctypes.c_int
    fn.argtypes = [PF, ctypes.c_int] + [ctypes.c_void_p] * 6 + [PF, PF, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    for c, rows in enumerate(steps_of_chains(CH)):
        run, cid, ratio = int(CH["chain_run"][c]), int(CH["chain_id"][c]), int(CH["chain_ratio"][c])
        st, pv = CH["est0"][c].copy(), CH["pvst0"][c].copy()
        for s, r in enumerate(rows):
            core = np.ascontiguousarray(chain_core(run, cid, s))
            out = np.zeros(4096 if ratio == RATIO_4_1 else 2048, np.float32)
            h, f, sd = (np.ascontiguousarray(CH[k][r]) for k in ("header", "frame", "side"))
            rc = fn(core.ctypes.data_as(PF), ratio, vp(h), vp(f), vp(sd), vp(st), None, None, out.ctypes.data_as(PF), None, None,
                    vp(np.ascontiguousarray(CH["pvc_side"][r])), vp(pv))
            want = CH["crc"][r]
            assert rc == CH["ret"][r], (c, s)
            assert crc(out) == want[0], ("out", c, s, ratio)
            assert state_crc(st, h, f, CH["apply"][r], ratio) == want[2], ("state", c, s, ratio)
            assert crc(pv) == want[5], ("pvc state", c, s, ratio)


@pytest.mark.gpu
@pytest.mark.parametrize("ratio", [RATIO_8_3, RATIO_4_1])
def test_gpu_walks_the_reference_chains(ratio):
    import torch
    import libxaac_amd
    ctx = libxaac_amd.XaacContext(0, 0)
    dev = torch.device("cuda:0")
    order = steps_of_chains(CH)
    chains = [c for c in range(len(order)) if int(CH["chain_ratio"][c]) == ratio]
    assert len(chains) > 10
    t_st = torch.from_numpy(np.ascontiguousarray(CH["est0"][chains])).to(dev)
    t_pv = torch.from_numpy(np.ascontiguousarray(CH["pvst0"][chains])).to(dev)
    width = 4096 if ratio == RATIO_4_1 else 2048
    for s in range(max(len(order[c]) for c in chains)):
        act = [i for i, c in enumerate(chains) if s < len(order[c])]
        rows = [order[chains[i]][s] for i in act]
        m = len(act)
        idx = torch.tensor(act, device=dev)
        core = torch.from_numpy(np.stack([chain_core(int(CH["chain_run"][chains[i]]), int(CH["chain_id"][chains[i]]), s) for i in act])).to(dev)
        g = lambda k: torch.from_numpy(np.ascontiguousarray(CH[k][rows])).to(dev)
        st, pv = t_st[idx].contiguous(), t_pv[idx].contiguous()
        out = torch.zeros((m, width), dtype=torch.float32, device=dev)
        status = torch.full((m,), 7, dtype=torch.int32, device=dev)
        ws = torch.zeros(ctx.esbr_workspace_bytes(m, ratio), dtype=torch.uint8, device=dev)
        ctx.esbr_sbr_process_batch(core, g("header"), g("frame"), g("side"), st, out, ws, status, pvc_side=g("pvc_side"), pvc_state=pv,
                                   sbr_ratio=ratio)
        ctx.sync()
        t_st[idx], t_pv[idx] = st, pv
        assert np.array_equal(status.cpu().numpy(), CH["ret"][rows]), s
        o, stn, pvn = out.cpu().numpy(), st.cpu().numpy(), pv.cpu().numpy()
        for j, r in enumerate(rows):
            want = CH["crc"][r]
            assert crc(o[j]) == want[0], ("out", chains[act[j]], s)
            assert state_crc(stn[j], CH["header"][r], CH["frame"][r], CH["apply"][r], ratio) == want[2], ("state", chains[act[j]], s)
            assert crc(pvn[j]) == want[5], ("pvc state", chains[act[j]], s)
    ctx.close()


@pytest.mark.gpu
def test_gpu_on_fuzzed_grids_equals_the_oracle(oracle):
    """the first six frames of every chain with every second frame's envelope grid fuzzed (anything in 0..19: unsorted, empty,
    running past the frame's end) and the band limit moving -- side info no parser makes, which the boundary has to contain -- at
    8:3 and at 4:1 (where a border is four QMF rows and the matrices are the 80- / 82-row scratch): the GPU chain against the oracle
    frame by frame: return codes, output words, state checksums.  (tests/test_sbr_core_sanitized.py runs the same under ASan.)"""
    import torch
    import libxaac_amd
    import sbr_capture as cap
    from test_env_pairs_cpu import _fuzz_frame
    fn = oracle.lib.xo_esbr_sbr_frame_ratio
    fn.restype = ctypes.c_int
    fn.argtypes = [PF, ctypes.c_int] + [ctypes.c_void_p] * 6 + [PF, PF, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    ctx = libxaac_amd.XaacContext(0, 0)
    dev = torch.device("cuda:0")
    order = steps_of_chains(CH)
    rng = np.random.default_rng(int(os.environ.get("XAAC_FUZZ_SEED", "4")))
    taken = refused = 0
    for ratio in (RATIO_8_3, RATIO_4_1):
        chains = [c for c in range(len(order)) if int(CH["chain_ratio"][c]) == ratio and len(order[c]) >= 6]
        n, width = len(chains), 4096 if ratio == RATIO_4_1 else 2048
        st = [CH["est0"][c].copy() for c in chains]
        pv = [CH["pvst0"][c].copy() for c in chains]
        for s in range(6):
            hdr, frm, side, pvs, cores = [], [], [], [], []
            for i, c in enumerate(chains):
                r = order[c][s]
                h, f = np.ascontiguousarray(CH["header"][r]).copy(), np.ascontiguousarray(CH["frame"][r]).copy()
                hh, ff = cap.Header.from_buffer(h), cap.Frame.from_buffer(f)
                if s % 2 == 1:
                    _fuzz_frame(rng, hh, ff, (c + s) % 3)
                if s % 4 == 3:
                    ff.max_qmf_subband_aac = int(np.clip(ff.max_qmf_subband_aac + rng.integers(-6, 7), hh.sub_band_start, 32))
                hdr.append(h); frm.append(f)
                side.append(np.ascontiguousarray(CH["side"][r])); pvs.append(np.ascontiguousarray(CH["pvc_side"][r]))
                cores.append(np.ascontiguousarray(chain_core(int(CH["chain_run"][c]), int(CH["chain_id"][c]), s)))
            up = lambda rows: torch.from_numpy(np.stack(rows)).to(dev)
            t_st, t_pv = up(st), up(pv)
            out = torch.zeros((n, width), dtype=torch.float32, device=dev)
            status = torch.full((n,), 7, dtype=torch.int32, device=dev)
            ws = torch.zeros(ctx.esbr_workspace_bytes(n, ratio), dtype=torch.uint8, device=dev)
            ctx.esbr_sbr_process_batch(up(cores), up(hdr), up(frm), up(side), t_st, out, ws, status, pvc_side=up(pvs), pvc_state=t_pv,
                                       sbr_ratio=ratio)
            ctx.sync()
            g_rc, g_out, g_st, g_pv = status.cpu().numpy(), out.cpu().numpy(), t_st.cpu().numpy(), t_pv.cpu().numpy()
            for i, c in enumerate(chains):
                o = np.zeros(width, np.float32)
                rc = fn(cores[i].ctypes.data_as(PF), ratio, vp(hdr[i]), vp(frm[i]), vp(side[i]), vp(st[i]), None, None,
                        o.ctypes.data_as(PF), None, None, vp(pvs[i]), vp(pv[i]))
                assert g_rc[i] == rc, ("rc", c, s, int(g_rc[i]), rc)
                if rc != 0:        # a refused frame: the chain goes on from the device's states
                    st[i], pv[i] = g_st[i].copy(), g_pv[i].copy()
                    refused += 1
                    continue
                taken += 1
                apply = cap.Frame.from_buffer_copy(frm[i].tobytes()).apply_processing
                assert np.array_equal(g_out[i].view(np.uint32), o.view(np.uint32)), ("out", c, s, ratio)
                assert state_crc(g_st[i], hdr[i], frm[i], apply, ratio) == state_crc(st[i], hdr[i], frm[i], apply, ratio), ("state", c, s, ratio)
                assert crc(g_pv[i]) == crc(pv[i]), ("pvc state", c, s, ratio)
    ctx.close()
    assert taken > 150 and refused < taken
