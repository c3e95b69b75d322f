#!/usr/bin/env python3
"""the host front end alone (xaac_parse_batch_run) by thread count: frames/s for N copies of a committed stream"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_decoder
from libxaac_amd import decoder
name = sys.argv[1] if len(sys.argv) > 1 else "mix_aot29_32k"
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
data = open(os.path.join(ROOT, "tests", "golden", "streams", name + ".aac"), "rb").read()
for t in (1, 4, 16, 32, 48, 64, 96, 128, 256):
    if t > (os.cpu_count() or 1): break
    n = copies if t >= 16 else max(64, copies // 16)
    best = max(bench_decoder.parser_only(decoder, data, n, t) for _ in range(2))
    print("threads %3d  streams %5d  %.3e frames/s  (%.2e per thread)" % (t, n, best, best / t), flush=True)
