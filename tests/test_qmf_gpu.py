"""HIP SBR QMF banks through the C ABI vs the oracle: bit-exact outputs AND persistent state,
low-power and HQ modes, mono/stereo interleave, odd batch sizes, multi-frame chains."""
import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    import libxaac_amd
    c = libxaac_amd.XaacContext(0, 0)
    yield c
    c.close()


@pytest.mark.parametrize("low_pow", [1, 0])
@pytest.mark.parametrize("n_ch,ch_fac", [(1, 1), (6, 2), (37, 1), (128, 2)])
def test_analysis_chain(ctx, oracle, low_pow, n_ch, ch_fac):
    import torch
    import libxaac_amd
    rng = np.random.default_rng(100 + n_ch + low_pow)
    ss = 64 if low_pow else 128
    o_state = np.zeros((n_ch, libxaac_amd.QMF_ANA_STATE_WORDS), np.int16)
    t_state = torch.zeros((n_ch, libxaac_amd.QMF_ANA_STATE_WORDS), dtype=torch.int16, device="cuda")
    for f in range(7):   # > 5 frames: the ring phase has period 5 frames
        amp = 32768 if f != 2 else 100
        pcm = rng.integers(-amp, amp, n_ch * 1024).astype(np.int16)
        want, o_state = oracle_lib.qmf_analysis_batch(oracle, pcm, o_state, low_pow, 32, ss, ch_fac)
        qmf = torch.full((n_ch, 32, ss), 0x55555555, dtype=torch.int32, device="cuda")
        ctx.qmf_analysis_batch(torch.from_numpy(pcm).cuda(), t_state, qmf, low_pow, 32, ss, ch_fac)
        torch.cuda.synchronize()
        got = qmf.cpu().numpy()
        if low_pow:
            assert np.array_equal(got[:, :, :32], want[:, :, :32]), f
        else:
            assert np.array_equal(got[:, :, :32], want[:, :, :32]) and np.array_equal(got[:, :, 64:96], want[:, :, 64:96]), f
        assert np.array_equal(t_state.cpu().numpy(), o_state), "state after frame %d" % f


@pytest.mark.parametrize("low_pow", [1, 0])
@pytest.mark.parametrize("n_ch,ch_fac", [(1, 1), (6, 2), (37, 1), (128, 2)])
def test_synthesis_chain(ctx, oracle, low_pow, n_ch, ch_fac):
    import torch
    import libxaac_amd
    rng = np.random.default_rng(200 + n_ch + low_pow)
    ss = 64 if low_pow else 128
    o_state = np.zeros((n_ch, libxaac_amd.QMF_SYN_STATE_WORDS), np.int16)
    t_state = torch.zeros((n_ch, libxaac_amd.QMF_SYN_STATE_WORDS), dtype=torch.int16, device="cuda")
    for f in range(6):
        mag = int(rng.integers(10, 30))
        qmf = rng.integers(-(1 << mag), 1 << mag, (n_ch, 32, ss)).astype(np.int32)
        scale = np.stack([rng.integers(-12, 4, n_ch), rng.integers(-12, 4, n_ch), rng.integers(-12, 4, n_ch),
                          rng.integers(-8, 0, n_ch)], 1).astype(np.int16)
        lsb = int(rng.integers(8, 33)); usb = int(rng.integers(lsb, 65))
        want, o_state = oracle_lib.qmf_synthesis_batch(oracle, qmf, scale, o_state, low_pow, lsb, usb, 6, ch_fac)
        pcm = torch.zeros(n_ch * 2048, dtype=torch.int16, device="cuda")
        t_qmf = torch.from_numpy(qmf).cuda()
        before = t_qmf.clone()
        ctx.qmf_synthesis_batch(t_qmf, torch.from_numpy(scale).cuda(), t_state, pcm, low_pow, lsb, usb, 6, ss, ch_fac)
        torch.cuda.synchronize()
        assert torch.equal(t_qmf, before), "qmf input must not be modified"
        assert np.array_equal(pcm.cpu().numpy(), want), f
        assert np.array_equal(t_state.cpu().numpy(), o_state), "state after frame %d" % f


def test_analysis_then_synthesis_is_near_identity(ctx):
    """property at BASELINE batch size: 8192 streams x 2 channels of a sine through the HQ analysis bank and the
    64-band synthesis bank (upper 32 bands empty) reproduce the sine (upsampled x2) once the banks' delay has passed"""
    import torch
    import libxaac_amd
    n_ch = 16384
    t = np.arange(1024 * 4)
    sine = (8000 * np.sin(2 * np.pi * 440 * t / 24000)).astype(np.int16)
    a_state = torch.zeros((n_ch, libxaac_amd.QMF_ANA_STATE_WORDS), dtype=torch.int16, device="cuda")
    s_state = torch.zeros((n_ch, libxaac_amd.QMF_SYN_STATE_WORDS), dtype=torch.int16, device="cuda")
    qmf = torch.zeros((n_ch, 32, 128), dtype=torch.int32, device="cuda")
    scale = torch.tensor([[-8, -8, -8, -6]], dtype=torch.int16, device="cuda").repeat(n_ch, 1).contiguous()
    outs = []
    for f in range(4):
        pcm_in = torch.from_numpy(np.tile(sine[1024 * f:1024 * (f + 1)], n_ch)).cuda()
        ctx.qmf_analysis_batch(pcm_in, a_state, qmf, 0, 32, 128, 1)
        pcm = torch.zeros(n_ch * 2048, dtype=torch.int16, device="cuda")
        ctx.qmf_synthesis_batch(qmf, scale, s_state, pcm, 0, 32, 32, 6, 128, 1)
        torch.cuda.synchronize()
        p = pcm.view(n_ch, 2048)
        assert (p == p[0]).all(), "all streams carry the same signal"
        outs.append(p[0].cpu().numpy())
    y = np.concatenate(outs).astype(np.float64)
    tt = np.arange(len(y))
    # fit amplitude/phase of the 440 Hz tone at the doubled rate on the last two frames
    seg = slice(4096, 8192)
    basis = np.stack([np.sin(2 * np.pi * 440 * tt[seg] / 48000), np.cos(2 * np.pi * 440 * tt[seg] / 48000)], 1)
    coef, res, *_ = np.linalg.lstsq(basis, y[seg], rcond=None)
    amp = np.hypot(*coef)
    err = y[seg] - basis @ coef
    assert amp > 1000 and np.sqrt(np.mean(err ** 2)) < 0.02 * amp
