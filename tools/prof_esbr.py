#!/usr/bin/env python3
"""profiling aid: bench.py's Path A leg alone (secondary.c4_esbr, with and without the harmonic transposer), for
rocprofv3 --kernel-trace --stats -- python tools/prof_esbr.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import libxaac_amd
import bench
dev = torch.device("cuda:0")
ctx = libxaac_amd.XaacContext(0, torch.cuda.current_stream(dev).cuda_stream)
print(json.dumps(bench.secondary_esbr(torch, libxaac_amd, ctx, dev, int(os.environ.get("STEPS", "20")), 3,
                                      hip_streams=int(os.environ.get("HIP_STREAMS", "1")))))
