#!/usr/bin/env python3
"""Tiny workload for rocprofv3 --pmc passes over the SBR chains: a few C3 (HE-AACv1) and C4 (HE-AACv2) steps of
bench.py's inputs plus one torch copy of known size (the calibration point for FETCH_SIZE / WRITE_SIZE)."""
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
n3, n4 = bench.FRAMES_PER_STEP * bench.CH, bench.FRAMES_PER_STEP
b3 = bench.make_inputs_c3(torch, dev, 2, 0)
b4 = bench.make_inputs_c4(torch, dev, 2, 0)
ws3 = torch.zeros(ctx.sbr_lp_workspace_bytes(n3), dtype=torch.uint8, device=dev)
ws4 = torch.zeros(ctx.sbr_hq_workspace_bytes(n4, True), dtype=torch.uint8, device=dev)
a = torch.zeros(64 * 1024 * 1024, dtype=torch.int32, device=dev)   # 256 MiB
c = torch.empty_like(a)
for i in range(6):
    b = b3[i % 2]
    ctx.imdct_process_batch(b["spec"], b["ics"], b["overlap"], b["state"], None, b["core_pcm"], None, ch_fac=1,
                            pcm_mode=libxaac_amd.PCM_SBR)
    ctx.sbr_lp_process_batch(b["core_pcm"], b["hdr"], b["frames"][i % 4], b["sbr_state"], b["pcm"], ws3, None,
                             in_ch_fac=1, out_ch_fac=2)
for i in range(6):
    b = b4[i % 2]
    ctx.imdct_process_batch(b["spec"], b["ics"], b["overlap"], b["state"], None, b["core_pcm"], None, ch_fac=1,
                            pcm_mode=libxaac_amd.PCM_SBR)
    fr, pfr = b["frames"][i % 4]
    ctx.sbr_hq_process_batch(b["core_pcm"], b["hdr"], fr, b["sbr_state"], b["pcm"], ws4, pfr, b["ps_state"])
c.copy_(a)   # calibration: reads 256 MiB, writes 256 MiB
torch.cuda.synchronize()
print("done")
