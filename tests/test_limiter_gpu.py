"""GPU parity of xaac_peak_limiter_process_batch (through the C ABI) with the oracle and with the reference's
recorded vectors: outputs, PCM16 and the whole state, bit for bit."""
import ctypes
import os

import numpy as np
import pytest

import limiter_cases as lc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "limiter_ref.npz")


@pytest.fixture(scope="module")
def ctx():
    import libxaac_amd
    c = libxaac_amd.XaacContext(0)
    yield c
    c.close()


def states_to_tensor(states, torch):
    raw = np.frombuffer(bytes(states), np.uint8).reshape(len(states), -1).copy()
    return torch.from_numpy(raw).cuda()


def tensor_to_states(t, n):
    raw = np.ascontiguousarray(t.cpu().numpy())
    arr = (lc.LimiterState * n)()
    ctypes.memmove(arr, raw.ctypes.data, raw.nbytes)
    return arr


def run_gpu(ctx, torch, x, q, states, nch, frame_len, stride=None):
    n = len(states)
    xs = torch.from_numpy(x.copy()).cuda()
    qs = torch.from_numpy(q.copy()).cuda()
    st = states_to_tensor(states, torch)
    pcm = torch.zeros(n * frame_len * nch, dtype=torch.int16, device="cuda")
    status = torch.full((n,), 77, dtype=torch.int32, device="cuda")
    ws = torch.zeros(ctx.peak_limiter_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    ctx.peak_limiter_process_batch(xs, qs, st, nch, ws, frame_len=frame_len, pcm16=pcm, stride=stride, status=status)
    ctx.sync()
    return xs.cpu().numpy(), pcm.cpu().numpy(), tensor_to_states(st, n), status.cpu().numpy()


def test_reference_vectors(ctx):
    import torch
    g = np.load(GOLD)
    for ci, (nch, rate, frame_len) in enumerate(g["chains"]):
        nch, frame_len = int(nch), int(frame_len)
        states = np.ascontiguousarray(g["state_%d" % ci])
        st = (lc.LimiterState * 1)()
        ctypes.memmove(st, states[0].ctypes.data, ctypes.sizeof(lc.LimiterState))
        for f in range(g["in_%d" % ci].shape[0]):
            out, pcm, st, status = run_gpu(ctx, torch, g["in_%d" % ci][f], g["q_%d" % ci][f], st, nch, frame_len)
            want = lc.LimiterState()
            ctypes.memmove(ctypes.byref(want), states[f + 1].ctypes.data, ctypes.sizeof(want))
            assert status[0] == 0
            assert np.array_equal(out, g["out_%d" % ci][f]), (ci, f)
            assert lc.state_view(st[0]) == lc.state_view(want), (ci, f)


@pytest.mark.parametrize("nch,rate,frame_len", [(2, 48000, 1024), (1, 44100, 1024), (2, 96000, 1024), (1, 8000, 1024),
                                                 (2, 32000, 777), (6, 48000, 1024), (2, 48000, 100)])
def test_chains_vs_oracle(ctx, oracle, nch, rate, frame_len):
    """every kind of signal in one batch, eight frames with the state carried on the GPU"""
    import torch
    init, _, batch = lc.bind(oracle.lib, "xo")
    n = 2 * len(lc.KINDS) + 3
    rng = np.random.default_rng(rate + nch)
    so = (lc.LimiterState * n)()
    for i in range(n):
        init(ctypes.byref(so[i]), nch, rate)
    sg = (lc.LimiterState * n)()
    ctypes.memmove(sg, so, ctypes.sizeof(so))
    stride = frame_len * nch + (8 if frame_len != 1024 else 0)
    for frame in range(8):
        x = np.zeros(n * stride, np.int32)
        for i in range(n):
            x[i * stride:i * stride + frame_len * nch] = lc.signal(rng, lc.KINDS[(i + 3 * frame) % len(lc.KINDS)], frame_len, nch)
        q = rng.integers(0, 3, n * nch).astype(np.int8) if frame == 5 else rng.integers(1, 3, n * nch).astype(np.int8)
        xo = x.copy()
        po = np.zeros(n * frame_len * nch, np.int16)
        batch(n, frame_len, nch, xo.ctypes.data_as(lc.P32), stride, q.ctypes.data_as(lc.P8), so, po.ctypes.data_as(lc.P16))
        xg, pg, sg, status = run_gpu(ctx, torch, x, q, sg, nch, frame_len, stride)
        assert not status.any()
        assert np.array_equal(xg, xo), frame
        assert np.array_equal(pg, po), frame
        for i in range(n):
            assert lc.state_view(sg[i]) == lc.state_view(so[i]), (frame, i)


def test_off_branch_and_misfit_state(ctx, oracle):
    import torch
    init, _, batch = lc.bind(oracle.lib, "xo")
    n, nch = 4, 2
    so = (lc.LimiterState * n)()
    for i in range(n):
        init(ctypes.byref(so[i]), nch, 48000)
    so[1].limiter_on = 0
    so[1].pre_smoothed_gain = 0.0
    so[2].num_channels = 1          # does not fit a 2-channel batch: flagged, left alone
    so[3].attack_time_samples = 481
    sg = (lc.LimiterState * n)()
    ctypes.memmove(sg, so, ctypes.sizeof(so))
    rng = np.random.default_rng(3)
    x = np.concatenate([lc.signal(rng, "loud", 1024, nch) for _ in range(n)])
    q = np.full(n * nch, 2, np.int8)
    xg, pg, sg2, status = run_gpu(ctx, torch, x, q, sg, nch, 1024)
    assert status.tolist() == [0, 0, -1, -1]
    xo = x.copy()
    po = np.zeros(n * 1024 * nch, np.int16)
    batch(2, 1024, nch, xo.ctypes.data_as(lc.P32), 1024 * nch, q.ctypes.data_as(lc.P8), so, po.ctypes.data_as(lc.P16))
    assert np.array_equal(xg, xo)   # streams 2, 3 untouched
    assert np.array_equal(pg[:2 * 2048], po[:2 * 2048])
    for i in range(2):
        assert lc.state_view(sg2[i]) == lc.state_view(so[i])
    for i in (2, 3):
        assert bytes(sg2[i]) == bytes(sg[i])


def test_state_with_stale_max_idx(ctx, oracle):
    """a state whose max_idx does not point at the window's maximum cannot come out of init + process, but the
    reference runs on it (it trusts the index): the kernel's per-sample walk has to give the same"""
    import torch
    init, _, batch = lc.bind(oracle.lib, "xo")
    n, nch = 10, 2
    rng = np.random.default_rng(21)
    so = (lc.LimiterState * n)()
    for i in range(n):
        init(ctypes.byref(so[i]), nch, 48000)
    sg = (lc.LimiterState * n)()
    for frame in range(5):
        if frame in (1, 3):
            for i in range(n):
                so[i].max_idx = (so[i].max_idx + 1 + 23 * i) % so[i].attack_time_samples
        ctypes.memmove(sg, so, ctypes.sizeof(so))
        x = np.concatenate([lc.signal(rng, lc.KINDS[(i + frame) % len(lc.KINDS)], 1024, nch) for i in range(n)])
        q = rng.integers(1, 3, n * nch).astype(np.int8)
        xo = x.copy()
        po = np.zeros(n * 1024 * nch, np.int16)
        batch(n, 1024, nch, xo.ctypes.data_as(lc.P32), 1024 * nch, q.ctypes.data_as(lc.P8), so, po.ctypes.data_as(lc.P16))
        xg, pg, sg, status = run_gpu(ctx, torch, x, q, sg, nch, 1024)
        assert np.array_equal(xg, xo) and np.array_equal(pg, po), frame
        for i in range(n):
            assert lc.state_view(sg[i]) == lc.state_view(so[i]), (frame, i)


def test_imdct_to_limiter_chain(ctx, oracle):
    """the AAC-LC tail as the decoder runs it: IMDCT out32 + qshift_adj straight into the limiter, on the GPU"""
    import torch
    init, _, batch = lc.bind(oracle.lib, "xo")
    n_streams, nch = 12, 2
    n = n_streams * nch
    rng = np.random.default_rng(11)
    so = (lc.LimiterState * n_streams)()
    for i in range(n_streams):
        init(ctypes.byref(so[i]), nch, 48000)
    st_g = states_to_tensor(so, torch)
    ws = torch.zeros(ctx.peak_limiter_workspace_bytes(n_streams), dtype=torch.uint8, device="cuda")
    ovl_o = np.zeros((n, 512), np.int32)
    state_o = np.zeros((n, 2), np.uint8)
    ovl_g = torch.zeros(n, 512, dtype=torch.int32, device="cuda")
    state_g = torch.zeros(n, 2, dtype=torch.uint8, device="cuda")
    for frame in range(4):
        spec = (rng.integers(-(1 << 28), 1 << 28, (n, 1024)) >> rng.integers(0, 12, (n, 1))).astype(np.int32)
        ics = np.stack([np.zeros(n, np.uint8), rng.integers(0, 2, n).astype(np.uint8)], 1)
        want = oracle.imdct_batch(spec, ics, ovl_o, state_o, ch_fac=nch)
        ovl_o, state_o = want["overlap"], want["state"]
        x = want["out32"].reshape(-1).copy()
        po = np.zeros(n * 1024, np.int16)
        batch(n_streams, 1024, nch, x.ctypes.data_as(lc.P32), 1024 * nch, want["qshift_adj"].ctypes.data_as(lc.P8), so,
              po.ctypes.data_as(lc.P16))
        out32 = torch.zeros(n * 1024, dtype=torch.int32, device="cuda")
        qadj = torch.zeros(n, dtype=torch.int8, device="cuda")
        pcm = torch.zeros(n * 1024, dtype=torch.int16, device="cuda")
        ctx.imdct_process_batch(torch.from_numpy(spec).cuda(), torch.from_numpy(ics).cuda(), ovl_g, state_g, out32=out32,
                                qshift_adj=qadj, ch_fac=nch)
        ctx.peak_limiter_process_batch(out32, qadj, st_g, nch, ws, pcm16=pcm)
        ctx.sync()
        assert np.array_equal(out32.cpu().numpy(), x), frame
        assert np.array_equal(pcm.cpu().numpy(), po), frame
    sg = tensor_to_states(st_g, n_streams)
    for i in range(n_streams):
        assert lc.state_view(sg[i]) == lc.state_view(so[i])


def test_full_size_batch_and_empty_batch(ctx, oracle):
    """8192 stereo streams (BASELINE's batch): a handful of distinct signals tiled over the batch, three frames with the
    state on the GPU; every copy must equal the oracle's result for its signal.  Then the empty batch and bad arguments."""
    import torch
    import libxaac_amd
    init, _, batch = lc.bind(oracle.lib, "xo")
    n, m, nch = 8192, 2 * len(lc.KINDS), 2
    rng = np.random.default_rng(77)
    idx = (np.arange(n) * 3 + np.arange(n) // m) % m
    so = (lc.LimiterState * m)()
    for i in range(m):
        init(ctypes.byref(so[i]), nch, 48000)
    st = states_to_tensor(so, torch)[torch.from_numpy(idx).cuda()].contiguous()
    ws = torch.zeros(ctx.peak_limiter_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    pcm = torch.zeros(n * 2048, dtype=torch.int16, device="cuda")
    for frame in range(3):
        x = np.stack([lc.signal(rng, lc.KINDS[(i + frame) % len(lc.KINDS)], 1024, nch) for i in range(m)])
        q = rng.integers(1, 3, (m, nch)).astype(np.int8)
        xo, po = x.reshape(-1).copy(), np.zeros(m * 2048, np.int16)
        batch(m, 1024, nch, xo.ctypes.data_as(lc.P32), 2048, q.ctypes.data_as(lc.P8), so, po.ctypes.data_as(lc.P16))
        xg = torch.from_numpy(np.ascontiguousarray(x[idx]).reshape(-1)).cuda()
        ctx.peak_limiter_process_batch(xg, torch.from_numpy(np.ascontiguousarray(q[idx]).reshape(-1)).cuda(), st, nch, ws,
                                       pcm16=pcm)
        ctx.sync()
        assert np.array_equal(xg.cpu().numpy().reshape(n, 2048), xo.reshape(m, 2048)[idx]), frame
        assert np.array_equal(pcm.cpu().numpy().reshape(n, 2048), po.reshape(m, 2048)[idx]), frame
    got = st.cpu().numpy()
    for i in list(range(0, n, 641)) + [n - 1]:
        g = lc.LimiterState.from_buffer_copy(got[i].tobytes())
        assert lc.state_view(g) == lc.state_view(so[idx[i]]), i
    # nothing to do / nothing sensible to do
    empty = torch.zeros(0, dtype=torch.int32, device="cuda")
    ctx.peak_limiter_process_batch(empty, torch.zeros(0, dtype=torch.int8, device="cuda"),
                                   torch.zeros((0, libxaac_amd.LIMITER_STATE_BYTES), dtype=torch.uint8, device="cuda"), nch, ws)
    with pytest.raises(libxaac_amd.XaacError):
        ctx.peak_limiter_process_batch(xg, torch.zeros(n * 9, dtype=torch.int8, device="cuda"), st, 9, ws, stride=2048)
    with pytest.raises(libxaac_amd.XaacError):
        ctx.peak_limiter_process_batch(xg, torch.zeros(n * nch, dtype=torch.int8, device="cuda"), st, nch, ws[:1000])


@pytest.mark.parametrize("nch,frame_len", [(2, 1024), (1, 1024), (3, 600)])
def test_planar_block_layout(ctx, oracle, nch, frame_len):
    """planar = 1: the WORD32 block is [channel][frame_len] (what the IMDCT writes with ch_fac = 1); same results,
    PCM16 interleaved as always"""
    import torch
    init, _, batch = lc.bind(oracle.lib, "xo")
    n = 2 * len(lc.KINDS)
    rng = np.random.default_rng(5 + nch)
    so = (lc.LimiterState * n)()
    for i in range(n):
        init(ctypes.byref(so[i]), nch, 48000)
    st = states_to_tensor(so, torch)
    ws = torch.zeros(ctx.peak_limiter_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    pcm = torch.zeros(n * frame_len * nch, dtype=torch.int16, device="cuda")
    for frame in range(4):
        x = np.stack([lc.signal(rng, lc.KINDS[(i + frame) % len(lc.KINDS)], frame_len, nch) for i in range(n)])
        q = rng.integers(1, 3, (n, nch)).astype(np.int8)
        xo, po = x.reshape(-1).copy(), np.zeros(n * frame_len * nch, np.int16)
        batch(n, frame_len, nch, xo.ctypes.data_as(lc.P32), frame_len * nch, q.ctypes.data_as(lc.P8), so, po.ctypes.data_as(lc.P16))
        xp = np.ascontiguousarray(x.reshape(n, frame_len, nch).transpose(0, 2, 1))      # [stream][channel][sample]
        xg = torch.from_numpy(xp.reshape(-1)).cuda()
        ctx.peak_limiter_process_batch(xg, torch.from_numpy(q.reshape(-1)).cuda(), st, nch, ws, frame_len=frame_len, pcm16=pcm,
                                       planar=True)
        ctx.sync()
        got = xg.cpu().numpy().reshape(n, nch, frame_len).transpose(0, 2, 1).reshape(-1)
        assert np.array_equal(got, xo), frame
        assert np.array_equal(pcm.cpu().numpy(), po), frame
    sg = tensor_to_states(st, n)
    for i in range(n):
        assert lc.state_view(sg[i]) == lc.state_view(so[i]), i
