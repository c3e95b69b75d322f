#!/usr/bin/env python3
"""Time xaac_pvc_process_batch on 8192 channels (GPU box): average launch duration from HIP events on the context's stream."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import libxaac_amd  # noqa: E402
import pvc_structs as ps  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    items = [ps.chain(100 + k, 1)[0] for k in range(256)]
    rep = n // 256
    frame = torch.from_numpy(np.stack([np.frombuffer(bytes(i[0]), np.uint8) for i in items])).repeat(rep, 1).to(dev)
    re = torch.from_numpy(np.stack([i[1] for i in items])).repeat(rep, 1, 1).to(dev)
    im = torch.from_numpy(np.stack([i[2] for i in items])).repeat(rep, 1, 1).to(dev)
    state = torch.zeros((n, libxaac_amd.PVC_STATE_BYTES), dtype=torch.uint8, device=dev)
    out = torch.zeros((n, 16, 64), dtype=torch.float32, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    for _ in range(5):
        ctx.pvc_process_batch(frame, re, im, state, out, status)
    ctx.sync()
    t = ctx.time_launches(lambda: ctx.pvc_process_batch(frame, re, im, state, out, status), 50) if hasattr(ctx, "time_launches") else None
    if t is None:
        import time
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(50):
            ctx.pvc_process_batch(frame, re, im, state, out, status)
        ctx.sync()
        t = (time.perf_counter() - t0) / 50 * 1e3
    read = n * (32 * 16 * 2 * 4 + 40 + 188)     # the low bands' rows a frame touches (2:1: 32 rows x at most 12 of 64 floats, as 64 B sectors) + frame + state
    wrote = n * (16 * 64 * 4 + 188)
    print("xaac_pvc_process_batch: %d channels, %.1f us per launch, %.0f GB/s of algorithmic bytes (%.1f MB)" % (n, t * 1e3, (read + wrote) / t / 1e6, (read + wrote) / 1e6))
    assert not status.cpu().numpy().any()
