"""xaac_imdct960_process_batch (frame_length 960) on the MI355X against the oracle restatement, which
tests/test_imdct960_oracle_vs_reference.py pins on the compiled reference: all 16 (previous, current) sequence pairs x
window shapes x levels in one batch, both PCM hand-offs, stereo interleave, a legal walk with overlap and state carried on
the device, refused window bytes, a full-size batch."""
import ctypes

import numpy as np
import pytest

P32 = ctypes.POINTER(ctypes.c_int32)
P16 = ctypes.POINTER(ctypes.c_int16)
P8 = ctypes.POINTER(ctypes.c_int8)
PU8 = ctypes.POINTER(ctypes.c_uint8)
LEGAL_NEXT = {0: (0, 1), 1: (2, 3), 2: (2, 3), 3: (0, 1)}


def oracle_batch(oracle, spec, ovl, state, ics, pcm_mode):
    fn = oracle.lib.xo_imdct960_batch
    fn.restype = None
    fn.argtypes = [ctypes.c_int, P32, P32, PU8, PU8, P32, P16, P8, ctypes.c_int]
    n = spec.shape[0]
    out32, pcm, qadj = np.zeros((n, 960), np.int32), np.zeros((n, 960), np.int16), np.zeros(n, np.int8)
    fn(n, spec.ctypes.data_as(P32), ovl.ctypes.data_as(P32), state.ctypes.data_as(PU8), ics.ctypes.data_as(PU8),
       out32.ctypes.data_as(P32), pcm.ctypes.data_as(P16), qadj.ctypes.data_as(P8), pcm_mode)
    return out32, pcm, qadj


def spectra(rng, n, levels):
    spec = np.zeros((n, 960), np.int32)
    for c in range(n):
        lv = int(levels[c % len(levels)])
        if c % 3 == 1:
            idx = rng.integers(0, 960, 8)
            spec[c, idx] = rng.integers(-lv, lv + 1, 8)
        else:
            spec[c] = rng.integers(-lv, lv + 1, 960)
    return spec


@pytest.mark.gpu
@pytest.mark.parametrize("pcm_mode", [0, 1])
def test_every_transition_vs_oracle(oracle, pcm_mode):
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    rng = np.random.default_rng(96 + pcm_mode)
    combos = [(ps, s, pw, w) for ps in range(4) for s in range(4) for pw in range(2) for w in range(2)]
    levels = (0, 1, 300, 2 ** 17, 2 ** 24, 2 ** 30, 2 ** 31 - 1)
    n = len(combos) * len(levels) + 3   # not a multiple of four: the last workgroup is partly empty
    ics = np.zeros((n, 2), np.uint8)
    state = np.zeros((n, 2), np.uint8)
    for c in range(n):
        ps, s, pw, w = combos[(c // len(levels)) % len(combos)]
        ics[c], state[c] = (s, w), (ps, pw)
    spec = spectra(rng, n, levels)
    spec[5::14] = -2 ** 31
    ovl = rng.integers(-2 ** 17, 2 ** 17, (n, 480)).astype(np.int32)
    ho, hs = ovl.copy(), state.copy()
    want32, want16, wantq = oracle_batch(oracle, spec, ho, hs, ics, pcm_mode)
    d_ovl, d_state = torch.from_numpy(ovl.copy()).to(dev), torch.from_numpy(state.copy()).to(dev)
    out32 = torch.zeros(n * 960, dtype=torch.int32, device=dev)
    pcm = torch.zeros(n * 960, dtype=torch.int16, device=dev)
    qadj = torch.zeros(n, dtype=torch.int8, device=dev)
    status = torch.full((n,), 9, dtype=torch.int32, device=dev)
    d_spec = torch.from_numpy(spec).to(dev)
    ctx.imdct960_process_batch(d_spec, torch.from_numpy(ics).to(dev), d_ovl, d_state, out32, pcm, qadj, 1, pcm_mode, status)
    ctx.sync()
    assert np.array_equal(d_spec.cpu().numpy(), spec)
    assert status.cpu().tolist() == [0] * n
    g32 = out32.cpu().numpy().reshape(n, 960)
    bad = np.nonzero((g32 != want32).any(axis=1))[0]
    assert bad.size == 0, (bad[:5], [combos[(c // len(levels)) % len(combos)] for c in bad[:5]])
    assert np.array_equal(pcm.cpu().numpy().reshape(n, 960), want16)
    assert np.array_equal(qadj.cpu().numpy(), wantq)
    assert np.array_equal(d_ovl.cpu().numpy(), ho) and np.array_equal(d_state.cpu().numpy(), hs)


@pytest.mark.gpu
def test_stereo_walk_with_state_on_device(oracle):
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    rng = np.random.default_rng(961)
    n = 64   # 32 stereo access units
    ho, hs = np.zeros((n, 480), np.int32), np.zeros((n, 2), np.uint8)
    d_ovl, d_state = torch.zeros((n, 480), dtype=torch.int32, device=dev), torch.zeros((n, 2), dtype=torch.uint8, device=dev)
    seq = np.zeros(n, np.int64)
    for frame in range(24):
        seq = np.array([rng.choice(LEGAL_NEXT[int(s)]) for s in seq])
        ics = np.stack([seq, rng.integers(0, 2, n)], axis=1).astype(np.uint8)
        spec = spectra(rng, n, [int(2 ** rng.integers(6, 31)) for _ in range(5)])
        want32, _, wantq = oracle_batch(oracle, spec, ho, hs, ics, 0)
        out32 = torch.zeros(n * 960, dtype=torch.int32, device=dev)
        qadj = torch.zeros(n, dtype=torch.int8, device=dev)
        ctx.imdct960_process_batch(torch.from_numpy(spec).to(dev), torch.from_numpy(ics).to(dev), d_ovl, d_state, out32, None, qadj, 2)
        ctx.sync()
        got = out32.cpu().numpy().reshape(n // 2, 960, 2)   # interleaved access units
        assert np.array_equal(got[:, :, 0], want32[0::2]) and np.array_equal(got[:, :, 1], want32[1::2]), frame
        assert np.array_equal(qadj.cpu().numpy(), wantq)
        assert np.array_equal(d_ovl.cpu().numpy(), ho) and np.array_equal(d_state.cpu().numpy(), hs), frame


@pytest.mark.gpu
def test_refused_window_bytes_and_full_batch(oracle):
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    rng = np.random.default_rng(962)
    n = 16384
    spec = rng.integers(-2 ** 17, 2 ** 17, (n, 960)).astype(np.int32)
    spec[:, 640:] = 0
    ics = np.stack([np.zeros(n), np.arange(n) % 2], axis=1).astype(np.uint8)
    ics[7] = (4, 0)
    ics[8] = (0, 2)
    state = np.zeros((n, 2), np.uint8)
    state[9] = (5, 0)
    ovl = rng.integers(-2 ** 15, 2 ** 15, (n, 480)).astype(np.int32)
    ho, hs = ovl.copy(), state.copy()
    want32, _, _ = oracle_batch(oracle, spec, ho, hs, ics, 0)
    d_ovl, d_state = torch.from_numpy(ovl).to(dev), torch.from_numpy(state).to(dev)
    out32 = torch.full((n * 960,), 3, dtype=torch.int32, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.imdct960_process_batch(torch.from_numpy(spec).to(dev), torch.from_numpy(ics).to(dev), d_ovl, d_state, out32, None, None, 1, 0, status)
    ctx.sync()
    st = status.cpu().numpy()
    assert set(np.nonzero(st)[0].tolist()) == {7, 8, 9} and np.all(st[[7, 8, 9]] == libxaac_amd.BAD_WINDOW_SEQ)
    got = out32.cpu().numpy().reshape(n, 960)
    ok = np.ones(n, bool)
    ok[[7, 8, 9]] = False
    assert np.array_equal(got[ok], want32[ok]) and np.all(got[~ok] == 3)
    assert np.array_equal(d_ovl.cpu().numpy(), ho) and np.array_equal(d_state.cpu().numpy(), hs)


@pytest.mark.gpu
def test_reference_made_chains():
    """tests/golden/imdct960_ref.npz: 32 chains x 40 frames made by the compiled reference (tools/make_golden_imdct960.py),
    all chains as one batch per frame step with overlap and state on the device: every frame's output and overlap CRC"""
    import os
    import sys
    import torch
    import libxaac_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from make_golden_imdct960 import CHAINS, FRAMES, chain_spec, crc
    gold = np.load(os.path.join(root, "tests", "golden", "imdct960_ref.npz"))
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    d_ovl = torch.zeros((CHAINS, 480), dtype=torch.int32, device=dev)
    d_state = torch.zeros((CHAINS, 2), dtype=torch.uint8, device=dev)
    for f in range(FRAMES):
        spec = np.stack([chain_spec(c, f) for c in range(CHAINS)])
        ics = np.ascontiguousarray(gold["side"][:, f, :2]).astype(np.uint8)
        out32 = torch.zeros(CHAINS * 960, dtype=torch.int32, device=dev)
        qadj = torch.zeros(CHAINS, dtype=torch.int8, device=dev)
        ctx.imdct960_process_batch(torch.from_numpy(spec).to(dev), torch.from_numpy(ics).to(dev), d_ovl, d_state, out32, None, qadj)
        ctx.sync()
        got, ov = out32.cpu().numpy().reshape(CHAINS, 960), d_ovl.cpu().numpy()
        assert np.array_equal(qadj.cpu().numpy(), gold["side"][:, f, 2])
        for c in range(CHAINS):
            assert (crc(got[c]), crc(ov[c])) == tuple(int(v) for v in gold["crc"][c, f]), (c, f)
    assert np.array_equal(got, gold["last"][:, 0]) and np.array_equal(ov, gold["last"][:, 1, :480])
