"""HE-AACv2 on the GPU through the C ABI (HQ SBR + parametric stereo, xaac_sbr_hq_process_batch): against the
committed records of the real reference, and against the oracle on fuzzed chains with the SBR and PS state living
on the device."""
import ctypes
import os

import numpy as np
import pytest

import sbr_capture as cap

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P16 = ctypes.POINTER(ctypes.c_int16)
GOLDEN = os.path.join(ROOT, "tests", "golden", "sbr_hq_ps_records.bin.gz")


@pytest.fixture(scope="module")
def ctx():
    import libxaac_amd
    c = libxaac_amd.XaacContext(0, 0)
    yield c
    c.close()


def gpu_run(ctx, headers, frames, states, ps_frames, ps_states, pcm_in, max_band_hint=0):
    import torch
    n = len(states)
    t = lambda objs: torch.from_numpy(np.frombuffer(b"".join(bytes(o) for o in objs), np.uint8).reshape(n, -1).copy()).cuda()
    t_h, t_f, t_s = t(headers), t(frames), t(states)
    with_ps = ps_frames is not None
    t_pf, t_ps = (t(ps_frames), t(ps_states)) if with_ps else (None, None)
    out = torch.zeros(n * 2048 * (2 if with_ps else 1), dtype=torch.int16, device="cuda")
    status = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    ws = torch.zeros(ctx.sbr_hq_workspace_bytes(n, with_ps), dtype=torch.uint8, device="cuda")
    ctx.sbr_hq_process_batch(torch.from_numpy(np.ascontiguousarray(pcm_in)).cuda(), t_h, t_f, t_s, out, ws, t_pf, t_ps,
                             status, max_band_hint=max_band_hint)
    torch.cuda.synchronize()
    return out.cpu().numpy(), t_s.cpu().numpy(), (t_ps.cpu().numpy() if with_ps else None), status.cpu().numpy()


def test_reference_records(ctx):
    recs = cap.read_records(GOLDEN)
    pcm_in = np.concatenate([r["pcm_in"] for r in recs])
    out, st, ps, status = gpu_run(ctx, [r["header"] for r in recs], [r["frame"] for r in recs], [r["st0"] for r in recs],
                                  [r["ps_frame"] for r in recs], [r["ps0"] for r in recs], pcm_in)
    for i, r in enumerate(recs):
        assert status[i] == r["ret"]
        o = out[4096 * i:4096 * (i + 1)]
        assert np.array_equal(o[0::2], r["pcm_out"][0]), ("left", i, r["call"], int(np.sum(o[0::2] != r["pcm_out"][0])))
        assert np.array_equal(o[1::2], r["pcm_out"][1]), ("right", i, r["call"], int(np.sum(o[1::2] != r["pcm_out"][1])))
        got = cap.State.from_buffer_copy(st[i].tobytes())
        assert not cap.diff_state(got, r["st1"]), (i, r["call"], cap.diff_state(got, r["st1"])[:3])
        gps = cap.PsState.from_buffer_copy(ps[i].tobytes())
        assert not cap.diff_state(gps, r["ps1"]), (i, r["call"], cap.diff_state(gps, r["ps1"])[:3])


def _fuzz(rng, h, f, pf):
    for k in range(h.num_if_bands):
        f.sbr_invf_mode[k] = int(rng.integers(0, 4))
    h.limiter_gains = int(rng.integers(0, 4))
    h.interpol_freq = int(rng.integers(0, 2))
    h.smoothing_mode = int(rng.integers(0, 2))
    if rng.integers(0, 3) == 0:
        for k in range(h.num_sf_bands[1]):
            f.add_harmonics[k] = int(rng.integers(0, 4) == 0)
    if pf is not None:
        pf.iid_quant = int(rng.integers(0, 2))
        nenv = int(rng.integers(1, 5))
        borders = [0] + sorted(rng.choice(np.arange(1, 32), nenv - 1, replace=False).tolist()) + [32]
        for e in range(7):
            pf.border_position[e] = borders[e] if e < len(borders) else 0
        lim = 15 if pf.iid_quant else 7
        for e in range(nenv):
            for b in range(20):
                pf.iid_par_table[e][b] = int(rng.integers(-lim, lim + 1))
                pf.icc_par_table[e][b] = int(rng.integers(0, 8))


def test_fuzzed_chain_vs_oracle(ctx, oracle):
    """48 streams built from the golden records: 10 frames of fuzzed SBR / PS side info and random core PCM, the
    state carried on the device; every frame must equal the oracle.  One stream per step is left unprocessed
    (apply_processing = 0): its right channel and PS state must stay untouched."""
    recs = cap.read_records(GOLDEN)
    n = len(recs)
    rng = np.random.default_rng(19)
    states = [cap.State.from_buffer_copy(bytes(r["st0"])) for r in recs]
    pstates = [cap.PsState.from_buffer_copy(bytes(r["ps0"])) for r in recs]
    for step in range(10):
        headers, frames, pframes = [], [], []
        for i, r in enumerate(recs):
            h = cap.Header.from_buffer_copy(bytes(r["header"]))
            f = cap.Frame.from_buffer_copy(bytes(r["frame"]))
            pf = cap.PsFrame.from_buffer_copy(bytes(r["ps_frame"]))
            _fuzz(rng, h, f, pf)
            if i == step:
                f.apply_processing = 0
            headers.append(h); frames.append(f); pframes.append(pf)
        amp = [30000, 3000, 200][step % 3]
        pcm = rng.integers(-amp, amp, (n, 1024)).astype(np.int16)
        out, st_bytes, ps_bytes, status = gpu_run(ctx, headers, frames, states, pframes, pstates, pcm.reshape(-1))
        new_states, new_ps = [], []
        for i in range(n):
            so = cap.State.from_buffer_copy(bytes(states[i]))
            po = cap.PsState.from_buffer_copy(bytes(pstates[i]))
            ref_out = np.zeros(4096, np.int16)
            rc = oracle.lib.xo_sbr_dec_hq(ctypes.byref(headers[i]), ctypes.byref(frames[i]), ctypes.byref(so),
                                          ctypes.byref(pframes[i]), ctypes.byref(po), pcm[i].ctypes.data_as(P16), 1,
                                          ref_out.ctypes.data_as(P16), 2)
            o = out[4096 * i:4096 * (i + 1)]
            assert status[i] == rc, (step, i)
            assert np.array_equal(o[0::2], ref_out[0::2]), ("left", step, i, int(np.sum(o[0::2] != ref_out[0::2])))
            if frames[i].apply_processing:
                assert np.array_equal(o[1::2], ref_out[1::2]), ("right", step, i)
            else:
                assert not o[1::2].any()      # nothing written (the output tensor starts zeroed)
            gs = cap.State.from_buffer_copy(st_bytes[i].tobytes())
            gp = cap.PsState.from_buffer_copy(ps_bytes[i].tobytes())
            assert not cap.diff_state(gs, so), (step, i, cap.diff_state(gs, so)[:3])
            assert not cap.diff_state(gp, po), (step, i, cap.diff_state(gp, po)[:3])
            new_states.append(gs); new_ps.append(gp)
        states, pstates = new_states, new_ps


def test_hq_mono_without_ps_vs_oracle(ctx, oracle):
    """HQ SBR alone (HE-AAC mono in HQ mode): the same records run with channel_mode = mono and no PS buffers"""
    recs = cap.read_records(GOLDEN)[:24]
    n = len(recs)
    rng = np.random.default_rng(23)
    headers, frames, states = [], [], []
    for r in recs:
        h = cap.Header.from_buffer_copy(bytes(r["header"]))
        f = cap.Frame.from_buffer_copy(bytes(r["frame"]))
        _fuzz(rng, h, f, None)
        h.channel_mode = 1
        headers.append(h); frames.append(f); states.append(cap.State.from_buffer_copy(bytes(r["st0"])))
    pcm = rng.integers(-20000, 20000, (n, 1024)).astype(np.int16)
    out, st_bytes, _, status = gpu_run(ctx, headers, frames, states, None, None, pcm.reshape(-1))
    for i in range(n):
        so = cap.State.from_buffer_copy(bytes(states[i]))
        ref_out = np.zeros(2048, np.int16)
        rc = oracle.lib.xo_sbr_dec_hq(ctypes.byref(headers[i]), ctypes.byref(frames[i]), ctypes.byref(so), None, None,
                                      pcm[i].ctypes.data_as(P16), 1, ref_out.ctypes.data_as(P16), 1)
        assert status[i] == rc
        assert np.array_equal(out[2048 * i:2048 * (i + 1)], ref_out), (i, int(np.sum(out[2048 * i:2048 * (i + 1)] != ref_out)))
        gs = cap.State.from_buffer_copy(st_bytes[i].tobytes())
        assert not cap.diff_state(gs, so), (i, cap.diff_state(gs, so)[:3])


def test_full_size_batch_matches_reference_records(ctx):
    """BASELINE's batch size (8192 HE-AACv2 streams): the golden records tiled over the whole batch; every copy must
    come out exactly as the reference's record (both channels, SBR and PS state)"""
    import torch
    recs = cap.read_records(GOLDEN)
    m, n = len(recs), 8192
    idx = (np.arange(n) * 5 + np.arange(n) // m) % m
    row = lambda key: np.stack([np.frombuffer(bytes(r[key]), np.uint8) for r in recs])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a[idx])).cuda()
    t_h, t_f, t_s, t_pf, t_ps = t(row("header")), t(row("frame")), t(row("st0")), t(row("ps_frame")), t(row("ps0"))
    pcm_in = torch.from_numpy(np.ascontiguousarray(np.stack([r["pcm_in"] for r in recs])[idx]).reshape(-1)).cuda()
    out = torch.zeros(n * 4096, dtype=torch.int16, device="cuda")
    status = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    ws = torch.zeros(ctx.sbr_hq_workspace_bytes(n, True), dtype=torch.uint8, device="cuda")
    ctx.sbr_hq_process_batch(pcm_in, t_h, t_f, t_s, out, ws, t_pf, t_ps, status)
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(n, 2048, 2)
    left = np.stack([r["pcm_out"][0] for r in recs])[idx]
    right = np.stack([r["pcm_out"][1] for r in recs])[idx]
    assert np.array_equal(got[:, :, 0], left) and np.array_equal(got[:, :, 1], right)
    assert np.array_equal(status.cpu().numpy(), np.array([r["ret"] for r in recs], np.int32)[idx])
    got_st, got_ps = t_s.cpu().numpy(), t_ps.cpu().numpy()
    for i in list(range(0, n, 499)) + [n - 1]:
        r = recs[idx[i]]
        assert not cap.diff_state(cap.State.from_buffer_copy(got_st[i].tobytes()), r["st1"]), i
        assert not cap.diff_state(cap.PsState.from_buffer_copy(got_ps[i].tobytes()), r["ps1"]), i


def test_malformed_side_info_does_not_disturb_the_batch(ctx, oracle):
    """HE-AACv2 batch with one stream's SBR frame and another's PS frame filled with random bytes: the SBR one is
    refused (status -1); the PS one has its indices clamped into the tables (xp_frame_sanitize, the same code in the
    oracle), is reported with status -1 and decodes exactly as the oracle does; every other stream of the batch is
    bit-exact"""
    recs = cap.read_records(GOLDEN)[:16]
    rng = np.random.default_rng(8)
    frames = [cap.Frame.from_buffer_copy(bytes(r["frame"])) for r in recs]
    psf = [cap.PsFrame.from_buffer_copy(bytes(r["ps_frame"])) for r in recs]
    frames[3] = cap.Frame.from_buffer_copy(rng.integers(0, 256, ctypes.sizeof(cap.Frame), dtype=np.uint8).tobytes())
    frames[3].apply_processing = 1
    psf[11] = cap.PsFrame.from_buffer_copy(rng.integers(0, 256, ctypes.sizeof(cap.PsFrame), dtype=np.uint8).tobytes())
    pcm_in = np.concatenate([r["pcm_in"] for r in recs])
    out, st, ps, status = gpu_run(ctx, [r["header"] for r in recs], frames, [r["st0"] for r in recs], psf,
                                  [r["ps0"] for r in recs], pcm_in)
    assert status[3] == -1 and status[11] == -1
    r = recs[11]
    st = cap.State.from_buffer_copy(bytes(r["st0"])); pst = cap.PsState.from_buffer_copy(bytes(r["ps0"]))
    ref = np.zeros(4096, np.int16)
    rc = oracle.lib.xo_sbr_dec_hq(ctypes.byref(r["header"]), ctypes.byref(frames[11]), ctypes.byref(st),
                                  ctypes.byref(psf[11]), ctypes.byref(pst), r["pcm_in"].ctypes.data_as(P16), 1,
                                  ref.ctypes.data_as(P16), 2)
    assert rc == -1 and np.array_equal(out[4096 * 11:4096 * 12], ref), "clamped PS frame: GPU and oracle agree"
    assert not cap.diff_state(cap.PsState.from_buffer_copy(ps[11].tobytes()), pst)
    for i, r in enumerate(recs):
        if i in (3, 11):
            continue
        o = out[4096 * i:4096 * (i + 1)]
        assert status[i] == r["ret"]
        assert np.array_equal(o[0::2], r["pcm_out"][0]) and np.array_equal(o[1::2], r["pcm_out"][1]), i
        assert not cap.diff_state(cap.PsState.from_buffer_copy(ps[i].tobytes()), r["ps1"]), i


def test_streams_outside_the_narrow_rows_take_the_list_kernel(ctx, oracle):
    """The HQ core keeps 48-band rows in LDS and sends streams that touch a higher band through a second, list-driven
    launch with the reference's 64-band rows (sbr_core_kernel.hip).  Every third stream here carries overlap words above
    band 48 (left there by a previous frame with a wider SBR range) and one more has its synthesis bank limit up there;
    all must still equal the oracle, frame after frame, next to streams that stay on the narrow path."""
    recs = cap.read_records(GOLDEN)
    n = len(recs)
    rng = np.random.default_rng(23)
    states = [cap.State.from_buffer_copy(bytes(r["st0"])) for r in recs]
    pstates = [cap.PsState.from_buffer_copy(bytes(r["ps0"])) for r in recs]
    for i in range(0, n, 3):
        ov = np.frombuffer(states[i], dtype=np.int32, count=6 * 128, offset=cap.State.overlap.offset).reshape(6, 2, 64)
        ov[:, :, 48 + (i % 16):] = rng.integers(-2000, 2000, ov[:, :, 48 + (i % 16):].shape)
    states[1].syn_usb = 52
    for step in range(3):
        headers = [cap.Header.from_buffer_copy(bytes(r["header"])) for r in recs]
        frames = [cap.Frame.from_buffer_copy(bytes(r["frame"])) for r in recs]
        pframes = [cap.PsFrame.from_buffer_copy(bytes(r["ps_frame"])) for r in recs]
        pcm = rng.integers(-9000, 9000, (n, 1024)).astype(np.int16)
        out, st_bytes, ps_bytes, status = gpu_run(ctx, headers, frames, states, pframes, pstates, pcm.reshape(-1))
        for i in range(n):
            ref_out = np.zeros(4096, np.int16)
            rc = oracle.lib.xo_sbr_dec_hq(ctypes.byref(headers[i]), ctypes.byref(frames[i]), ctypes.byref(states[i]),
                                          ctypes.byref(pframes[i]), ctypes.byref(pstates[i]), pcm[i].ctypes.data_as(P16), 1,
                                          ref_out.ctypes.data_as(P16), 2)
            assert status[i] == rc, (step, i)
            assert np.array_equal(out[4096 * i:4096 * (i + 1)], ref_out), (step, i)
            gs = cap.State.from_buffer_copy(st_bytes[i].tobytes())
            gp = cap.PsState.from_buffer_copy(ps_bytes[i].tobytes())
            assert not cap.diff_state(gs, states[i]), (step, i, cap.diff_state(gs, states[i])[:3])
            assert not cap.diff_state(gp, pstates[i]), (step, i, cap.diff_state(gp, pstates[i])[:3])


def test_max_band_hint_leaves_the_list_launch_out_and_refuses_what_it_does_not_cover(ctx):
    """xaac_sbr_hq_batch.max_band_hint = 48: the caller's knowledge that no stream reaches above band 48.  The reference records
    (24 kHz cores) decode to the same words with and without it; a stream that does reach higher -- a synthesis bank limit at band
    52 here, overlap content above band 48 there -- is refused with XAAC_FATAL_BAD_ARG instead of taking the 64-band pass, and the
    streams beside it are untouched by that."""
    recs = cap.read_records(GOLDEN)
    n = len(recs)
    pcm_in = np.concatenate([r["pcm_in"] for r in recs])
    args = lambda states: ([r["header"] for r in recs], [r["frame"] for r in recs], states, [r["ps_frame"] for r in recs],
                           [r["ps0"] for r in recs], pcm_in)
    plain = gpu_run(ctx, *args([r["st0"] for r in recs]))
    hinted = gpu_run(ctx, *args([r["st0"] for r in recs]), max_band_hint=48)
    assert not hinted[3].any() or np.array_equal(hinted[3], plain[3])
    for a, b in zip(plain, hinted):
        assert np.array_equal(a, b)
    states = [cap.State.from_buffer_copy(bytes(r["st0"])) for r in recs]
    states[1].syn_usb = 52
    ov = np.frombuffer(states[4], dtype=np.int32, count=6 * 128, offset=cap.State.overlap.offset).reshape(6, 2, 64)
    ov[:, :, 50:] = 77
    wide = gpu_run(ctx, *args(states))                       # no hint: both streams go through the list launch
    out, st, ps, status = gpu_run(ctx, *args(states), max_band_hint=48)
    bad_arg = np.int32(np.uint32(0xFFFF8001))                # XAAC_FATAL_BAD_ARG
    assert status[1] == status[4] == bad_arg and wide[3][1] == recs[1]["ret"]
    for i in range(n):
        if i in (1, 4):
            continue
        assert status[i] == recs[i]["ret"]
        assert np.array_equal(out[4096 * i:4096 * (i + 1)], wide[0][4096 * i:4096 * (i + 1)]), i
        assert np.array_equal(st[i], wide[1][i]) and np.array_equal(ps[i], wide[2][i]), i


def test_fuzzed_envelope_grids_and_moving_band_limit_vs_oracle(ctx, oracle):
    """the HQ + PS chain on what decides between the core's two-envelope passes and its one-envelope chain and on what moves
    its rows about: envelope borders of every kind (parser-like grids, variable frames, anything in 0..19: unsorted, empty,
    behind slot 32, grids that end before slot 16), frequency resolutions that differ inside a pair, noise-floor rows that
    switch between two envelopes, transient envelopes, and a band limit that moves (xs_rescale_x_overlap's branches, rows
    wider than the narrow instantiation holds).  Ten frames with the state on the device, every frame against the oracle."""
    from test_env_pairs_cpu import _fuzz_frame
    recs = cap.read_records(GOLDEN)
    n = len(recs)
    rng = np.random.default_rng(88)
    states = [cap.State.from_buffer_copy(bytes(r["st0"])) for r in recs]
    pstates = [cap.PsState.from_buffer_copy(bytes(r["ps0"])) for r in recs]
    hdrs = [cap.Header.from_buffer_copy(bytes(r["header"])) for r in recs]
    pframes = [cap.PsFrame.from_buffer_copy(bytes(r["ps_frame"])) for r in recs]
    taken = refused = 0
    for step in range(10):
        frames = []
        for i, r in enumerate(recs):
            f = cap.Frame.from_buffer_copy(bytes(r["frame"]))
            _fuzz_frame(rng, hdrs[i], f, (i + step) % 3)
            if step % 4 == 3:
                f.max_qmf_subband_aac = int(np.clip(f.max_qmf_subband_aac + rng.integers(-6, 7), hdrs[i].sub_band_start, 32))
            frames.append(f)
        if step == 6:
            for h in hdrs:
                h.smoothing_mode = 1 - h.smoothing_mode
        amp = [30000, 3000, 200, 12, 0][step % 5]
        pcm = rng.integers(-amp, amp + 1, (n, 1024)).astype(np.int16)
        out, st_bytes, ps_bytes, status = gpu_run(ctx, hdrs, frames, states, pframes, pstates, pcm.reshape(-1))
        new_states, new_ps = [], []
        for i in range(n):
            so = cap.State.from_buffer_copy(bytes(states[i]))
            po = cap.PsState.from_buffer_copy(bytes(pstates[i]))
            ref_out = np.zeros(4096, np.int16)
            rc = oracle.lib.xo_sbr_dec_hq(ctypes.byref(hdrs[i]), ctypes.byref(frames[i]), ctypes.byref(so),
                                          ctypes.byref(pframes[i]), ctypes.byref(po), pcm[i].ctypes.data_as(P16), 1,
                                          ref_out.ctypes.data_as(P16), 2)
            assert status[i] == rc, (step, i, int(status[i]), rc)
            gs = cap.State.from_buffer_copy(st_bytes[i].tobytes())
            gp = cap.PsState.from_buffer_copy(ps_bytes[i].tobytes())
            if rc == 0:
                assert np.array_equal(out[4096 * i:4096 * (i + 1)], ref_out), ("pcm", step, i)
                assert not cap.diff_state(gs, so), (step, i, cap.diff_state(gs, so)[:3])
                assert not cap.diff_state(gp, po), (step, i, cap.diff_state(gp, po)[:3])
                new_states.append(so); new_ps.append(po)
                taken += 1
            else:          # a refused frame leaves the oracle's states half-written: the chain goes on from the GPU's
                new_states.append(gs); new_ps.append(gp)
                refused += 1
        states, pstates = new_states, new_ps
    assert taken > 250 and refused < taken
