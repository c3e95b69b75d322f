#!/usr/bin/env python3
"""Secondary measurement: the SBR QMF banks alone on a batch of 16384
channels (8192 stereo HE-AAC streams), HIP events on the launch stream.  Prints one JSON line per kernel.
Algorithmic bytes per channel-frame: analysis R 2048 (PCM16) + 644 state, W 644 state + 4096 (LP) / 8192 (HQ);
synthesis R 8192 (LP) / 16384 (HQ) + 2564 state + 8 scale, W 2564 state + 4096 PCM."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libxaac_amd  # noqa: E402

N = 16384
STEPS = int(os.environ.get("STEPS", "30"))
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx = libxaac_amd.XaacContext(0, stream.cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(1)


def timeit(fn):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(STEPS)]
    for a, b in ev:
        a.record(stream); fn(); b.record(stream)
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / STEPS


for low_pow in (1, 0):
    ss = 64 if low_pow else 128
    pcm = torch.randint(-20000, 20000, (N * 1024,), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
    a_state = torch.zeros((N, libxaac_amd.QMF_ANA_STATE_WORDS), dtype=torch.int16, device=dev)
    qmf = torch.zeros((N, 32, ss), dtype=torch.int32, device=dev)
    ms = timeit(lambda: ctx.qmf_analysis_batch(pcm, a_state, qmf, low_pow, 32, ss, 2))
    alg = N * (2048 + 2 * 644 + (4096 if low_pow else 8192))
    print(json.dumps({"kernel": "qmf_analysis_%s" % ("lp" if low_pow else "hq"), "channels": N, "ms": round(ms, 4),
                      "channel_frames_per_s": round(N / ms * 1e3), "alg_GBps": round(alg / ms / 1e6, 1),
                      "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4), "launch": ctx.last_launch()}))
    s_state = torch.zeros((N, libxaac_amd.QMF_SYN_STATE_WORDS), dtype=torch.int16, device=dev)
    scale = torch.tensor([[-8, -8, -8, -6]], dtype=torch.int16, device=dev).repeat(N, 1).contiguous()
    out = torch.zeros(N * 2048, dtype=torch.int16, device=dev)
    ms = timeit(lambda: ctx.qmf_synthesis_batch(qmf, scale, s_state, out, low_pow, 20, 40, 6, ss, 2))
    alg = N * ((8192 if low_pow else 16384) + 2 * 2564 + 8 + 4096)
    print(json.dumps({"kernel": "qmf_synthesis_%s" % ("lp" if low_pow else "hq"), "channels": N, "ms": round(ms, 4),
                      "channel_frames_per_s": round(N / ms * 1e3), "alg_GBps": round(alg / ms / 1e6, 1),
                      "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4), "launch": ctx.last_launch()}))
