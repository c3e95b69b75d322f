"""xaac_imdct_ld_process_batch (AAC-LD / AAC-ELD, 512 / 480 lines) on the MI355X: against the oracle restatement
(tests/test_imdct_ld_oracle_vs_reference.py pins it on the compiled reference) on batches with every level and both
shapes, stereo interleave, state carried on the device; against the reference-made chains of tests/golden/imdct_ld_ref.npz;
refused shape bytes."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
P32 = ctypes.POINTER(ctypes.c_int32)
P16 = ctypes.POINTER(ctypes.c_int16)


def oracle_fn(oracle):
    of = oracle.lib.xo_imdct_ld_process
    of.restype = ctypes.c_int
    of.argtypes = [P32, P32, P16, ctypes.c_int, ctypes.c_int, ctypes.c_int, P16, ctypes.c_int]
    return of


@pytest.mark.gpu
@pytest.mark.parametrize("frame_length", [512, 480])
@pytest.mark.parametrize("eld", [0, 1])
def test_stereo_chains_vs_oracle(oracle, frame_length, eld):
    import torch
    import libxaac_amd
    of = oracle_fn(oracle)
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    rng = np.random.default_rng(5 * frame_length + eld)
    n = 70   # 35 stereo units; not a multiple of four waves
    n_ov = 3 * frame_length if eld else frame_length // 2
    ho, hs = np.zeros((n, n_ov), np.int32), np.zeros(n, np.int16)
    d_ov = torch.zeros((n, n_ov), dtype=torch.int32, device=dev)
    d_sp = torch.zeros(n, dtype=torch.uint8, device=dev)
    levels = (0, 1, 300, 2 ** 17, 2 ** 24, 2 ** 30, 2 ** 31 - 1)
    for frame in range(10):
        spec = np.zeros((n, frame_length), np.int32)
        for c in range(n):
            lv = levels[(c + frame) % 7]
            spec[c] = rng.integers(-lv, lv + 1, frame_length)
            if (c + frame) % 5 == 1:
                spec[c, rng.integers(0, frame_length, frame_length - 8)] = 0
        spec[3] = -2 ** 31
        shape = rng.integers(0, 2, n).astype(np.uint8)
        want = np.zeros((n, frame_length), np.int16)
        for c in range(n):
            sp = hs[c:c + 1]
            assert of(spec[c].ctypes.data_as(P32), ho[c].ctypes.data_as(P32), sp.ctypes.data_as(P16), int(shape[c]), frame_length, eld,
                      want[c].ctypes.data_as(P16), 1) == -2
        pcm = torch.zeros(n * frame_length, dtype=torch.int16, device=dev)
        status = torch.full((n,), 9, dtype=torch.int32, device=dev)
        ctx.imdct_ld_process_batch(torch.from_numpy(spec).to(dev), torch.from_numpy(shape).to(dev), d_ov, d_sp, pcm, frame_length, eld, 2, status)
        ctx.sync()
        got = pcm.cpu().numpy().reshape(n // 2, frame_length, 2)
        assert status.cpu().tolist() == [0] * n
        assert np.array_equal(got[:, :, 0], want[0::2]) and np.array_equal(got[:, :, 1], want[1::2]), frame
        assert np.array_equal(d_ov.cpu().numpy(), ho), frame
        assert np.array_equal(d_sp.cpu().numpy(), hs.astype(np.uint8))


@pytest.mark.gpu
def test_reference_made_chains():
    import torch
    import libxaac_amd
    from make_golden_imdct_ld import CHAINS, CONFIGS, FRAMES, chain_spec, crc, n_overlap
    gold = np.load(os.path.join(ROOT, "tests", "golden", "imdct_ld_ref.npz"))
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    for g, (fl, eld) in enumerate(CONFIGS):
        d_ov = torch.zeros((CHAINS, n_overlap(fl, eld)), dtype=torch.int32, device=dev)
        d_sp = torch.zeros(CHAINS, dtype=torch.uint8, device=dev)
        for f in range(FRAMES):
            spec = np.stack([chain_spec(g, c, f) for c in range(CHAINS)])
            pcm = torch.zeros(CHAINS * fl, dtype=torch.int16, device=dev)
            ctx.imdct_ld_process_batch(torch.from_numpy(spec).to(dev), torch.from_numpy(np.ascontiguousarray(gold["shape"][g, :, f])).to(dev),
                                       d_ov, d_sp, pcm, fl, eld)
            ctx.sync()
            got, ov = pcm.cpu().numpy().reshape(CHAINS, fl), d_ov.cpu().numpy()
            for c in range(CHAINS):
                assert (crc(got[c]), crc(ov[c])) == tuple(int(v) for v in gold["crc"][g, c, f]), (fl, eld, c, f)
        assert np.array_equal(got, gold["last"][g, :, :fl])


@pytest.mark.gpu
def test_refused_shape_bytes_and_bad_arguments():
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = 8
    spec = torch.ones((n, 512), dtype=torch.int32, device=dev)
    shape = torch.tensor([0, 1, 2, 0, 0, 0, 0, 1], dtype=torch.uint8, device=dev)
    sp = torch.tensor([0, 0, 0, 3, 0, 0, 0, 0], dtype=torch.uint8, device=dev)
    ov = torch.full((n, 256), 5, dtype=torch.int32, device=dev)
    pcm = torch.full((n * 512,), 7, dtype=torch.int16, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.imdct_ld_process_batch(spec, shape, ov, sp, pcm, 512, 0, 1, status)
    ctx.sync()
    st = status.cpu().numpy()
    assert st[2] == st[3] == libxaac_amd.BAD_WINDOW_SEQ and (np.delete(st, [2, 3]) == 0).all()
    assert (pcm.cpu().numpy().reshape(n, 512)[[2, 3]] == 7).all() and (ov.cpu().numpy()[[2, 3]] == 5).all()
    assert sp.cpu().tolist() == [0, 1, 0, 3, 0, 0, 0, 1]
    b = libxaac_amd._ImdctLdBatch()
    b.n_ch, b.ch_fac, b.frame_length, b.eld = n, 1, 500, 0
    b.spec, b.window_shape, b.overlap, b.shape_prev, b.pcm16 = spec.data_ptr(), shape.data_ptr(), ov.data_ptr(), sp.data_ptr(), pcm.data_ptr()
    assert libxaac_amd.load_library().xaac_imdct_ld_process_batch(ctx._h, ctypes.byref(b)) != 0   # neither 512 nor 480
