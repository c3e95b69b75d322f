#!/usr/bin/env python3
"""Developer tool: time the HE-AACv2 chain (xaac_sbr_hq_process_batch, 8192 streams) for library variants built with
extra flags:  python tools/time_c4_variants.py tag1 "flags1" tag2 "flags2" ...   (run on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(tag, flags):
    src = os.path.join(ROOT, "libxaac_amd", "csrc")
    out = os.path.join(ROOT, "libxaac_amd", "libxaac_amd_%s.so" % tag)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-x", "hip"]
                          + flags + [os.path.join(src, f) for f in ("imdct_kernel.hip", "sbr_qmf_kernel.hip", "sbr_core_kernel.hip", "sbr_ps_kernel.hip", "limiter_kernel.hip", "xaac_abi.cpp")] + ["-o", out],
                          stderr=subprocess.DEVNULL)
    return out


def run(tag, lib):
    code = r'''
import os, sys, torch
sys.path.insert(0, %r)
import libxaac_amd
libxaac_amd.library_path = lambda: %r
import bench
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = libxaac_amd.XaacContext(0, stream.cuda_stream)
job = bench.Workload("c4", torch, libxaac_amd, ctx, dev, stream, 2, 0)
el, k = job.run(30, 4, torch.cuda.synchronize)
print(%r, "ms/step %%.4f" %% k, "check", job.verify(), job.refused())
''' % (ROOT, lib, tag)
    subprocess.run([sys.executable, "-c", code], stderr=subprocess.DEVNULL)


if __name__ == "__main__":
    a = sys.argv[1:]
    for i in range(0, len(a), 2):
        run(a[i], build(a[i], a[i + 1].split()))
