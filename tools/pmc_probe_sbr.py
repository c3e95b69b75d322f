#!/usr/bin/env python3
"""Tiny workload for rocprofv3 --pmc passes over the C4 chain (HE-AACv2): eight steps of bench.py's inputs plus one torch
copy of known size (the calibration point for FETCH_SIZE / WRITE_SIZE)."""
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
n4 = bench.FRAMES_PER_STEP
b4 = bench.make_inputs_c4(torch, dev, 2, 0)
ws4 = torch.zeros(ctx.sbr_hq_workspace_bytes(n4, True), dtype=torch.uint8, device=dev)
a = torch.zeros(64 * 1024 * 1024, dtype=torch.int32, device=dev)   # 256 MiB
c = torch.empty_like(a)
for i in range(8):
    b = b4[i % 2]
    ctx.imdct_process_batch(b["spec"], b["ics"], b["overlap"], b["state"], None, b["core_pcm"], None, ch_fac=1,
                            pcm_mode=libxaac_amd.PCM_SBR)
    fr, pfr = b["frames"][i % 4]
    ctx.sbr_hq_process_batch(b["core_pcm"], b["hdr"], fr, b["sbr_state"], b["pcm"], ws4, pfr, b["ps_state"])
c.copy_(a)   # calibration: reads 256 MiB, writes 256 MiB
torch.cuda.synchronize()
print("done")
