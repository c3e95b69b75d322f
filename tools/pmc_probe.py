#!/usr/bin/env python3
"""Tiny workload for rocprofv3 --pmc passes: a few IMDCT launches on the C2
batch plus one torch copy of known size (the calibration point for
FETCH_SIZE / WRITE_SIZE, see MI355X_MICROARCH.md §HBM)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import libxaac_amd  # noqa: E402

dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx = libxaac_amd.XaacContext(0, stream.cuda_stream)
batches = bench.make_inputs(torch, dev, 4, 0)
a = torch.zeros(64 * 1024 * 1024, dtype=torch.int32, device=dev)   # 256 MiB
b = torch.empty_like(a)
for i in range(12):
    x = batches[i % 4]
    ctx.imdct_process_batch(x["spec"], x["ics"], x["overlap"], x["state"], None, x["pcm"], None, ch_fac=2)
b.copy_(a)   # calibration: reads 256 MiB, writes 256 MiB
torch.cuda.synchronize()
print("done")
