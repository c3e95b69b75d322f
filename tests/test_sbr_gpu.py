"""Low-power SBR (HE-AACv1 channel-frames) on the GPU through the C ABI: against the committed records of the
real reference, and against the oracle on long fuzzed chains with the state living on the device."""
import ctypes
import os

import numpy as np
import pytest

import sbr_capture as cap

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P16 = ctypes.POINTER(ctypes.c_int16)


@pytest.fixture(scope="module")
def ctx():
    import libxaac_amd
    c = libxaac_amd.XaacContext(0, 0)
    yield c
    c.close()


def gpu_run(ctx, headers, frames, states, pcm_in, in_ch_fac=1, out_ch_fac=1):
    """lists of ctypes structs (+ pcm_in int16[n*1024]) -> (pcm_out, new states as bytes rows, status)"""
    import torch
    import libxaac_amd
    n = len(states)
    t = lambda objs: torch.from_numpy(np.frombuffer(b"".join(bytes(o) for o in objs), np.uint8).reshape(n, -1).copy()).cuda()
    t_h, t_f, t_s = t(headers), t(frames), t(states)
    out = torch.zeros(n * 2048, dtype=torch.int16, device="cuda")
    status = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    ws = torch.zeros(ctx.sbr_lp_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    ctx.sbr_lp_process_batch(torch.from_numpy(np.ascontiguousarray(pcm_in)).cuda(), t_h, t_f, t_s, out, ws, status,
                             in_ch_fac, out_ch_fac)
    torch.cuda.synchronize()
    return out.cpu().numpy(), t_s.cpu().numpy(), status.cpu().numpy()


def test_reference_records(ctx):
    recs = cap.read_records(os.path.join(ROOT, "tests", "golden", "sbr_lp_records.bin.gz"))
    n = len(recs)
    pcm_in = np.concatenate([r["pcm_in"] for r in recs])
    out, st, status = gpu_run(ctx, [r["header"] for r in recs], [r["frame"] for r in recs], [r["st0"] for r in recs],
                              pcm_in)
    for i, r in enumerate(recs):
        assert status[i] == r["ret"]
        assert np.array_equal(out[2048 * i:2048 * (i + 1)], r["pcm_out"][0]), ("pcm", i, r["call"])
        got = cap.State.from_buffer_copy(st[i].tobytes())
        assert not cap.diff_state(got, r["st1"]), (i, r["call"], cap.diff_state(got, r["st1"])[:3])


def test_fuzzed_chain_vs_oracle_stereo_interleaved(ctx, oracle):
    """a batch of 2 x 36 channels built from the golden records: 12 frames of fuzzed side info and random core PCM,
    interleaved stereo in and out, state carried on the device; every frame must equal the oracle"""
    recs = cap.read_records(os.path.join(ROOT, "tests", "golden", "sbr_lp_records.bin.gz"))
    recs = recs[:72 - 72 % 2]
    n = len(recs)
    rng = np.random.default_rng(9)
    states = [cap.State.from_buffer_copy(bytes(r["st0"])) for r in recs]
    for step in range(12):
        headers, frames = [], []
        for r in recs:
            h = cap.Header.from_buffer_copy(bytes(r["header"]))
            f = cap.Frame.from_buffer_copy(bytes(r["frame"]))
            for k in range(h.num_if_bands):
                f.sbr_invf_mode[k] = int(rng.integers(0, 4))
            h.limiter_gains = int(rng.integers(0, 4))
            h.interpol_freq = int(rng.integers(0, 2))
            if rng.integers(0, 3) == 0:
                for k in range(h.num_sf_bands[1]):
                    f.add_harmonics[k] = int(rng.integers(0, 4) == 0)
            headers.append(h); frames.append(f)
        amp = [30000, 3000, 200][step % 3]
        pcm_planar = rng.integers(-amp, amp, (n, 1024)).astype(np.int16)
        pcm_il = pcm_planar.reshape(n // 2, 2, 1024).transpose(0, 2, 1).reshape(-1)       # L R L R ...
        out, st_bytes, status = gpu_run(ctx, headers, frames, states, pcm_il, in_ch_fac=2, out_ch_fac=2)
        out_planar = out.reshape(n // 2, 2048, 2).transpose(0, 2, 1).reshape(n, 2048)
        new_states = []
        for i in range(n):
            st = cap.State.from_buffer_copy(bytes(states[i]))
            want = np.zeros(2048, np.int16)
            pin = np.ascontiguousarray(pcm_planar[i])
            rc = oracle.lib.xo_sbr_dec_lp(ctypes.byref(headers[i]), ctypes.byref(frames[i]), ctypes.byref(st),
                                          pin.ctypes.data_as(P16), 1, want.ctypes.data_as(P16), 1)
            assert status[i] == rc, (step, i)
            assert np.array_equal(out_planar[i], want), ("pcm", step, i)
            got = cap.State.from_buffer_copy(st_bytes[i].tobytes())
            assert not cap.diff_state(got, st), (step, i, cap.diff_state(got, st)[:3])
            new_states.append(st)
        states = new_states


def test_workspace_and_argument_checks(ctx):
    import torch
    import libxaac_amd
    z = lambda n, dt=torch.uint8: torch.zeros(n, dtype=dt, device="cuda")
    with pytest.raises(libxaac_amd.XaacError):     # workspace too small
        ctx.sbr_lp_process_batch(z(2048, torch.int16), z((2, 336)), z((2, 1072)), z((2, 7300)), z(4096, torch.int16),
                                 z(1000))
    assert ctx.sbr_lp_workspace_bytes(16384) >= 16384 * 40 * 64 * 4


def test_full_size_batch_matches_reference_records(ctx):
    """BASELINE's batch size (16384 channel-frames): the golden records tiled over the whole batch -- every copy, wherever
    it lands in the grid, must come out exactly as the reference's record (outputs, state, status)"""
    import torch
    recs = cap.read_records(os.path.join(ROOT, "tests", "golden", "sbr_lp_records.bin.gz"))
    m, n = len(recs), 16384
    idx = (np.arange(n) * 7 + np.arange(n) // m) % m          # a shuffled tiling, so neighbours differ
    row = lambda key: np.stack([np.frombuffer(bytes(r[key]), np.uint8) for r in recs])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a[idx])).cuda()
    t_h, t_f, t_s = t(row("header")), t(row("frame")), t(row("st0"))
    pcm_in = torch.from_numpy(np.ascontiguousarray(np.stack([r["pcm_in"] for r in recs])[idx]).reshape(-1)).cuda()
    out = torch.zeros(n * 2048, dtype=torch.int16, device="cuda")
    status = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    ws = torch.zeros(ctx.sbr_lp_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    ctx.sbr_lp_process_batch(pcm_in, t_h, t_f, t_s, out, ws, status)
    torch.cuda.synchronize()
    want_out = np.stack([r["pcm_out"][0] for r in recs])[idx]
    want_st = np.stack([np.frombuffer(bytes(r["st1"]), np.uint8) for r in recs])
    assert np.array_equal(out.cpu().numpy().reshape(n, 2048), want_out)
    assert np.array_equal(status.cpu().numpy(), np.array([r["ret"] for r in recs], np.int32)[idx])
    got_st = t_s.cpu().numpy()
    for i in list(range(0, n, 997)) + [n - 1]:               # the state compare knows which bytes are don't-care
        assert not cap.diff_state(cap.State.from_buffer_copy(got_st[i].tobytes()), recs[idx[i]]["st1"]), i
    # and all copies of one record end in the same state bytes
    first = {}
    for i in range(n):
        k = int(idx[i])
        if k in first:
            assert np.array_equal(got_st[i], got_st[first[k]]), (i, k)
        else:
            first[k] = i
    assert want_st.shape[0] == m


def test_side_info_outside_the_structs_capacity_is_refused(ctx):
    """a frame whose counts / band numbers would index past the boundary structs is answered with status -1 (the
    reference's parser never produces one; the boundary does not trust its caller), its neighbours in the batch are
    untouched by it"""
    recs = cap.read_records(os.path.join(ROOT, "tests", "golden", "sbr_lp_records.bin.gz"))[:24]
    rng = np.random.default_rng(4)
    headers = [cap.Header.from_buffer_copy(bytes(r["header"])) for r in recs]
    frames = [cap.Frame.from_buffer_copy(bytes(r["frame"])) for r in recs]
    broken = {1: "num_env", 5: "tbl_hi", 9: "patch", 13: "garbage header", 17: "garbage frame", 21: "num_sf"}
    frames[1].num_env = 9
    headers[5].freq_band_tbl_hi[2] = 400
    headers[9].num_patches = 6
    headers[9].patch[5].src_start_band = -7
    raw = bytearray(bytes(headers[13]))
    raw[4:] = rng.integers(0, 256, len(raw) - 4, dtype=np.uint8).tobytes()     # keeps the 16 x 2 frame grid
    headers[13] = cap.Header.from_buffer_copy(bytes(raw))
    frames[17] = cap.Frame.from_buffer_copy(rng.integers(0, 256, ctypes.sizeof(cap.Frame), dtype=np.uint8).tobytes())
    frames[17].apply_processing = 1
    headers[21].num_sf_bands[1] = 60
    pcm_in = np.concatenate([r["pcm_in"] for r in recs])
    out, st, status = gpu_run(ctx, headers, frames, [r["st0"] for r in recs], pcm_in)
    for i, r in enumerate(recs):
        if i in broken:
            assert status[i] == -1, (i, broken[i], status[i])
        else:
            assert status[i] == r["ret"]
            assert np.array_equal(out[2048 * i:2048 * (i + 1)], r["pcm_out"][0]), i
            assert not cap.diff_state(cap.State.from_buffer_copy(st[i].tobytes()), r["st1"]), i


def test_fuzzed_envelope_grids_and_moving_band_limit_vs_oracle(ctx, oracle):
    """what decides between the core's two-envelope passes and its one-envelope chain, fuzzed per frame and carried over ten
    frames on the device: envelope borders (parser-like grids, variable frames, anything in 0..19: unsorted, empty, behind
    slot 32), frequency resolutions that differ inside a pair, noise-floor rows that switch between two envelopes,
    transient envelopes, interpol_freq, and a band limit that moves off sub_band_start (skip != 0: the one-envelope chain).
    Every frame must equal the oracle (whose own paired / one-envelope arrangements are held against each other and against
    the reference in tests/test_env_pairs_cpu.py)."""
    from test_env_pairs_cpu import _fuzz_frame
    recs = cap.read_records(os.path.join(ROOT, "tests", "golden", "sbr_lp_records.bin.gz"))
    n = len(recs)
    rng = np.random.default_rng(77)
    states = [cap.State.from_buffer_copy(bytes(r["st0"])) for r in recs]
    hdrs = [cap.Header.from_buffer_copy(bytes(r["header"])) for r in recs]
    for i, h in enumerate(hdrs):
        h.interpol_freq = 1 if i % 3 else h.interpol_freq
    refused = taken = 0
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
        out, st_bytes, status = gpu_run(ctx, hdrs, frames, states, pcm.reshape(-1))
        new_states = []
        for i in range(n):
            st = cap.State.from_buffer_copy(bytes(states[i]))
            want = np.zeros(2048, np.int16)
            pin = np.ascontiguousarray(pcm[i])
            rc = oracle.lib.xo_sbr_dec_lp(ctypes.byref(hdrs[i]), ctypes.byref(frames[i]), ctypes.byref(st),
                                          pin.ctypes.data_as(P16), 1, want.ctypes.data_as(P16), 1)
            assert status[i] == rc, (step, i, int(status[i]), rc)
            if rc == 0:
                assert np.array_equal(out[2048 * i:2048 * (i + 1)], want), ("pcm", step, i)
                got = cap.State.from_buffer_copy(st_bytes[i].tobytes())
                assert not cap.diff_state(got, st), (step, i, cap.diff_state(got, st)[:3])
                new_states.append(st)
                taken += 1
            else:                      # a refused frame leaves the oracle's state half-written; the chain goes on from the GPU's
                new_states.append(cap.State.from_buffer_copy(st_bytes[i].tobytes()))
                refused += 1
        states = new_states
    assert taken > 400 and refused < taken
