#!/usr/bin/env python3
"""Developer tool: which spectral amplitude makes the IMDCT output exceed full scale after the qshift_adj scale (i.e. makes
the peak limiter work): prints the share of samples above 2^31 per amplitude.  Run on the GPU box."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libxaac_amd
ctx = libxaac_amd.XaacContext(0)
n = 256
for e in range(17, 29):
    g = torch.Generator(device='cuda'); g.manual_seed(e)
    spec = torch.randint(-(1 << e), 1 << e, (n, 1024), generator=g, device='cuda', dtype=torch.int32)
    spec[:, 640:] = 0
    ics = torch.zeros((n, 2), dtype=torch.uint8, device='cuda')
    ovl = torch.zeros((n, 512), dtype=torch.int32, device='cuda'); st = torch.zeros((n, 2), dtype=torch.uint8, device='cuda')
    out = torch.zeros(n * 1024, dtype=torch.int32, device='cuda'); q = torch.zeros(n, dtype=torch.int8, device='cuda')
    for _ in range(2):
        ctx.imdct_process_batch(spec, ics, ovl, st, out32=out, qshift_adj=q, ch_fac=2)
    ctx.sync()
    o = out.cpu().numpy().astype(np.float64).reshape(n // 2, 1024, 2) * (2.0 ** q.cpu().numpy().astype(np.float64).reshape(n // 2, 1, 2))
    print(e, "q", np.unique(q.cpu().numpy()), "max %.3g" % np.abs(o).max(), "frac over 2^31: %.4f" % (np.abs(o) > 2.0 ** 31).mean())
