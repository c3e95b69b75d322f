#!/usr/bin/env python3
"""Tiny workload for rocprofv3 --pmc passes over the C4 chain (HE-AACv2; WORKLOAD=c3: the C3 chain, HE-AACv1 stereo):
eight steps of bench.py's inputs plus one torch copy of known size (the calibration point for FETCH_SIZE / WRITE_SIZE)."""
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
W = os.environ.get("WORKLOAD", "c4")
job = bench.Workload(W, torch, libxaac_amd, ctx, dev, stream, 2, 0)
a = torch.zeros(64 * 1024 * 1024, dtype=torch.int32, device=dev)   # 256 MiB
c = torch.empty_like(a)
for i in range(8):
    job.step(i)
c.copy_(a)   # calibration: reads 256 MiB, writes 256 MiB
torch.cuda.synchronize()
print("done")
