#!/usr/bin/env python3
"""GPU box: bench.py's secondary_round6 alone (for rocprofv3 --kernel-trace --stats: profiles/r06_*_round6_kernel_stats.txt)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import libxaac_amd  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
ctx = libxaac_amd.XaacContext(0, torch.cuda.current_stream().cuda_stream)
print(json.dumps(bench.secondary_round6(torch, libxaac_amd, ctx, dev, steps=int(os.environ.get("STEPS", "20")))))
