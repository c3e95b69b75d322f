"""The PVC envelope decoder (include/xaac_pvc.h; decoder/ixheaacd_pred_vec_block.c:176 + ixheaacd_qmf_enrg_calc,
ixheaacd_sbr_dec.c:80): the oracle (oracle/oracle_pvc.cpp = libxaac_amd/csrc/pvc.h run by a team of one) against the compiled
reference's own functions (oracle/ref_pvc_adapter.c) and against the committed reference-made chains
(tests/golden/pvc_ref.npz, tools/make_golden_pvc.py); the kernel behind xaac_pvc_process_batch against both on the GPU.
Compared as raw words: all 1024 output floats and the whole carried state of every frame."""
import ctypes
import os

import numpy as np
import pytest

import pvc_structs as ps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "pvc_ref.npz"))


def test_struct_sizes():
    import libxaac_amd
    assert ctypes.sizeof(ps.PvcFrame) == libxaac_amd.PVC_FRAME_BYTES and ctypes.sizeof(ps.PvcState) == libxaac_amd.PVC_STATE_BYTES


def test_oracle_walks_the_reference_made_chains(oracle):
    g = gold()
    fn = ps.bind(oracle.lib, "xo_pvc_process")
    for k, seed in enumerate(g["seeds"]):
        res = ps.walk(fn, ps.chain(int(seed), int(g["frames"])))
        bad = np.argwhere(res != g["res"][k])
        assert bad.size == 0, (int(seed), bad[:3].tolist())


@pytest.mark.parametrize("seed", range(8))
def test_oracle_vs_reference(oracle, reference, seed):
    """fresh chains against the reference itself, words compared (not CRCs)"""
    fo, fr = ps.bind(oracle.lib, "xo_pvc_process"), ps.bind(reference.lib, "ref_pvc_process")
    so, sr = ps.PvcState(), ps.PvcState()
    restarts = 0
    for n, (f, re, im, clear) in enumerate(ps.chain(7000 + seed, 40)):
        if clear:
            so.prev_pvc_flg = sr.prev_pvc_flg = 0
        restarts += so.prev_pvc_flg == 0 or f.first_bnd_idx * f.pvc_rate != so.prev_first_bnd_idx * so.prev_pvc_rate
        oo, orr = np.full((16, 64), np.float32(3)), np.full((16, 64), np.float32(5))
        a = fo(ctypes.byref(f), re.ctypes.data_as(ps.PF), im.ctypes.data_as(ps.PF), ctypes.byref(so), oo.ctypes.data_as(ps.PF))
        b = fr(ctypes.byref(f), re.ctypes.data_as(ps.PF), im.ctypes.data_as(ps.PF), ctypes.byref(sr), orr.ctypes.data_as(ps.PF))
        assert (a, b) == (0, 0)
        bad = np.argwhere(oo.view(np.uint32) != orr.view(np.uint32))
        assert bad.size == 0, (n, f.pvc_mode, f.pvc_rate, f.first_bnd_idx, bad[:3].tolist(), oo[tuple(bad[0])], orr[tuple(bad[0])])
        assert bytes(so) == bytes(sr), n
        assert np.all(oo[:, :f.first_bnd_idx] == 0) and np.all(oo[:, f.first_bnd_idx:] > 0)
    assert 2 <= restarts < 40


def test_parameters_outside_the_tables_are_refused(oracle, reference):
    fo, fr = ps.bind(oracle.lib, "xo_pvc_process"), ps.bind(reference.lib, "ref_pvc_process")
    f, re, im, _ = ps.chain(1, 1)[0]
    for name, value in (("pvc_mode", 0), ("pvc_mode", 3), ("pvc_rate", 3), ("pvc_rate", 0), ("first_bnd_idx", 33), ("first_bnd_idx", -1),
                        ("first_pvc_timeslot", 16), ("pvc_id", 128), ("first_bnd_idx@4:1", 17), ("first_bnd_idx@4:1", 32)):
        g = ps.PvcFrame.from_buffer_copy(bytes(f))
        if name == "pvc_id":
            g.pvc_id[9] = value
        elif name == "first_bnd_idx@4:1":   # the reference's energy rows hold sub-bands 0..15 only at 4:1 (sbr_dec.c:83-107)
            g.pvc_rate, g.first_bnd_idx = 4, value
        else:
            setattr(g, name, value)
        st = ps.PvcState()
        st.prev_pvc_id = 77
        before = bytes(st)
        out = np.full((16, 64), np.float32(9))
        assert fo(ctypes.byref(g), re.ctypes.data_as(ps.PF), im.ctypes.data_as(ps.PF), ctypes.byref(st), out.ctypes.data_as(ps.PF)) == -1
        assert bytes(st) == before and np.all(out == 9)
        if name == "pvc_mode":   # the one case the reference itself answers (pred_vec_block.c:218)
            assert fr(ctypes.byref(g), re.ctypes.data_as(ps.PF), im.ctypes.data_as(ps.PF), ctypes.byref(st), out.ctypes.data_as(ps.PF)) == -1


def _gpu_walk(chains, n_frames):
    """all chains side by side through xaac_pvc_process_batch, state on the device -> uint32[n_chains, n_frames, 3]"""
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = len(chains)
    state = torch.zeros((n, libxaac_amd.PVC_STATE_BYTES), dtype=torch.uint8, device=dev)
    res = np.zeros((n, n_frames, 3), np.uint32)
    for fr in range(n_frames):
        items = [c[fr] for c in chains]
        frame = torch.from_numpy(np.stack([np.frombuffer(bytes(i[0]), np.uint8) for i in items])).to(dev)
        re = torch.from_numpy(np.stack([i[1] for i in items])).to(dev)
        im = torch.from_numpy(np.stack([i[2] for i in items])).to(dev)
        clear = [k for k, i in enumerate(items) if i[3]]
        if clear:
            state[clear, ps.PvcState.prev_pvc_flg.offset] = 0
        out = torch.full((n, 16, 64), -7.0, dtype=torch.float32, device=dev)
        status = torch.full((n,), 5, dtype=torch.int32, device=dev)
        ctx.pvc_process_batch(frame, re, im, state, out, status)
        ctx.sync()
        o, s, rc = out.cpu().numpy(), state.cpu().numpy(), status.cpu().numpy()
        for k in range(n):
            res[k, fr] = (int(rc[k]) & 0xffffffff, ps.crc(o[k]), ps.crc(s[k]))
    ctx.close()
    return res


@pytest.mark.gpu
def test_gpu_walks_the_reference_made_chains():
    g = gold()
    chains = [ps.chain(int(s), int(g["frames"])) for s in g["seeds"]]
    res = _gpu_walk(chains, int(g["frames"]))
    bad = np.argwhere(res != g["res"])
    assert bad.size == 0, (len(bad), bad[:4].tolist())


@pytest.mark.gpu
def test_gpu_batch_vs_oracle_and_refusals(oracle):
    """a 1031-channel batch of fresh chains against the oracle, with frames the kernel must refuse in between (status -1,
    output and state untouched)"""
    fn = ps.bind(oracle.lib, "xo_pvc_process")
    n, frames = 1031, 3
    chains = [ps.chain(9000 + k, frames) for k in range(n)]
    for k in range(0, n, 50):
        chains[k][1][0].pvc_rate = 3
        chains[k + 1][2][0].pvc_id[3] = 200
    want = np.stack([ps.walk(fn, c) for c in chains])
    assert (want[:, :, 0] != 0).sum() == 2 * len(range(0, n, 50))
    res = _gpu_walk(chains, frames)
    bad = np.argwhere(res != want)
    assert bad.size == 0, (len(bad), bad[:4].tolist())


@pytest.mark.gpu
def test_gpu_refuses_4_to_1_frames_on_32_row_buffers(oracle):
    """a pvc_rate 4 frame reads 64 QMF rows: in a batch whose channels are 32 rows apart it is refused (status -1, nothing
    written) instead of reading the next channel's rows -- or, for the last channel, past the allocation; the 2:1 frames of
    the same batch decode as before"""
    import torch
    import libxaac_amd
    fn = ps.bind(oracle.lib, "xo_pvc_process")
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    items = []
    seed = 100
    while len(items) < 24:
        f, re, im, _ = ps.chain(seed, 1)[0]
        seed += 1
        if len(items) % 4 == 3:              # every fourth channel a 4:1 frame; the last channel is one (it would leave the allocation)
            f.pvc_rate, f.first_bnd_idx = 4, min(int(f.first_bnd_idx), 16)
        else:
            f.pvc_rate = 2
        items.append((f, re, im))
    rates = np.array([i[0].pvc_rate for i in items])
    assert (rates == 4).sum() == 6 and rates[-1] == 4
    n = len(items)
    frame = torch.from_numpy(np.stack([np.frombuffer(bytes(i[0]), np.uint8) for i in items])).to(dev)
    re = torch.from_numpy(np.stack([i[1][:32] for i in items])).to(dev)      # [n, 32, 64]
    im = torch.from_numpy(np.stack([i[2][:32] for i in items])).to(dev)
    state = torch.zeros((n, libxaac_amd.PVC_STATE_BYTES), dtype=torch.uint8, device=dev)
    out = torch.full((n, 16, 64), -7.0, dtype=torch.float32, device=dev)
    status = torch.full((n,), 5, dtype=torch.int32, device=dev)
    ctx.pvc_process_batch(frame, re, im, state, out, status)
    ctx.sync()
    o, s, rc = out.cpu().numpy(), state.cpu().numpy(), status.cpu().numpy()
    assert np.array_equal(rc, np.where(rates == 4, -1, 0))
    assert np.all(o[rates == 4] == -7.0) and not s[rates == 4].any()
    for k in np.argwhere(rates == 2)[:, 0]:
        st = ps.PvcState()
        want = np.zeros((16, 64), np.float32)
        assert fn(ctypes.byref(items[k][0]), items[k][1].ctypes.data_as(ps.PF), items[k][2].ctypes.data_as(ps.PF), ctypes.byref(st),
                  want.ctypes.data_as(ps.PF)) == 0
        assert np.array_equal(o[k].view(np.uint32), want.view(np.uint32)) and bytes(s[k]) == bytes(st)
    ctx.close()
