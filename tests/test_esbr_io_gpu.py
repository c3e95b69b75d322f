"""The two hand-offs around the Path A branch (include/xaac_esbr.h: xaac_esbr_core_from_pcm16_batch,
xaac_esbr_pcm16_from_float_batch) against their definitions in numpy: the core's PCM16 as float planes
(decoder/ixheaacd_api.c:3385-3432) and ixheaacd_samples_sat (decoder/ixheaacd_decode_main.c:82-107: saturate to
[-32768, 32767], truncate towards zero) for mono-twice, PS and channel-pair outputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def samples_sat(x):
    return np.trunc(np.clip(x.astype(np.float64), -32768.0, 32767.0)).astype(np.int16)


def test_core_planes_and_pcm_out():
    import torch
    import libxaac_amd
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, torch.cuda.current_stream(dev).cuda_stream)
    rng = np.random.default_rng(5)
    for ch_fac, n in ((1, 7), (2, 5)):
        pcm = rng.integers(-32768, 32768, size=(n, 1024, ch_fac), dtype=np.int16)
        core = torch.zeros(n * ch_fac, 1024, dtype=torch.float32, device=dev)
        ctx.esbr_core_from_pcm16(torch.from_numpy(pcm.reshape(-1)).to(dev), core, ch_fac=ch_fac)
        want = pcm.transpose(0, 2, 1).reshape(n * ch_fac, 1024).astype(np.float32)
        assert np.array_equal(core.cpu().numpy(), want)
    n = 6
    x = (rng.standard_normal((2 * n, 2048)) * 20000.0).astype(np.float32)
    x[0, :8] = [32767.0, 32767.5, 32768.0, -32768.0, -32768.5, -32769.0, 0.999, -0.999]
    x[1, :4] = [3.0e38, -3.0e38, 1.5, -1.5]
    xd = torch.from_numpy(x).to(dev)
    out = torch.zeros(n * 4096, dtype=torch.int16, device=dev)
    # a pair: planes 2 i and 2 i + 1
    ctx.esbr_pcm16_from_float(xd, xd[1:], out, stride=4096)
    want = np.stack((samples_sat(x[0::2]), samples_sat(x[1::2])), axis=2)
    assert np.array_equal(out.cpu().numpy().reshape(n, 2048, 2), want)
    # PS: left and right arrays; mono: the same plane twice
    left, right = xd[:n].contiguous(), xd[n:].contiguous()
    ctx.esbr_pcm16_from_float(left, right, out)
    assert np.array_equal(out.cpu().numpy().reshape(n, 2048, 2), np.stack((samples_sat(x[:n]), samples_sat(x[n:])), axis=2))
    ctx.esbr_pcm16_from_float(left, left, out)
    assert np.array_equal(out.cpu().numpy().reshape(n, 2048, 2), np.repeat(samples_sat(x[:n])[:, :, None], 2, axis=2))
    ctx.close()
