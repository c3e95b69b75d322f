#!/usr/bin/env python3
"""A few launches of every f4 transform kernel (USAC FD, 960-line, AAC-LD / ELD 512 and 480) on resident batches: the command
tools/pmc_f4.sh runs under rocprofv3 for SQ counters and kernel times (bench.py: secondary_f4 is the workload)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import libxaac_amd
    import bench
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, torch.cuda.current_stream().cuda_stream)
    print(bench.secondary_f4(torch, libxaac_amd, ctx, dev, launches=6, hip_streams=int(os.environ.get('HIP_STREAMS', '1'))))


if __name__ == "__main__":
    main()
