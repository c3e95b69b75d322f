#!/usr/bin/env python3
"""A few launches of the 960-line IMDCT kernel (long frames, then short frames) on a resident batch: the command
tools/pmc_i960.sh runs under rocprofv3 for counters and kernel times."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import libxaac_amd
    n = 16384
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(0)
    spec = torch.from_numpy(rng.integers(-2 ** 17, 2 ** 17, (n, 960)).astype(np.int32)).to(dev)
    ov = torch.zeros((n, 480), dtype=torch.int32, device=dev)
    out = torch.zeros(n * 960, dtype=torch.int32, device=dev)
    for seq in (0, 2):
        ics = torch.tensor([[seq, 1]] * n, dtype=torch.uint8, device=dev)
        st = torch.tensor([[seq, 1]] * n, dtype=torch.uint8, device=dev)
        for _ in range(6):
            ctx.imdct960_process_batch(spec, ics, ov, st, out)
        ctx.sync()


if __name__ == "__main__":
    main()
