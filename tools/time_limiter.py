#!/usr/bin/env python3
"""Developer tool: time xaac_peak_limiter_process_batch (8192 stereo streams) for a library variant built with
extra -D flags, on quiet / mixed / loud signals; with -DXL_PROFILE also the kernel's phase cycles.
usage: time_limiter.py <tag> [hipcc flags...]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(tag, flags):
    src = os.path.join(ROOT, "libxaac_amd", "csrc")
    out = os.path.join(ROOT, "libxaac_amd", "libxaac_amd_%s.so" % tag)
    files = ("imdct_kernel.hip", "sbr_qmf_kernel.hip", "sbr_core_kernel.hip", "sbr_ps_kernel.hip", "limiter_kernel.hip",
             "xaac_abi.cpp")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-ffp-contract=off", "-x", "hip"] + flags + [os.path.join(src, f) for f in files] + ["-o", out])
    return out


def main():
    tag, flags = sys.argv[1], sys.argv[2:]
    import numpy as np
    import torch
    import libxaac_amd
    if flags:
        out = build(tag, flags)
        libxaac_amd.library_path = lambda: out
    dev = torch.device("cuda:0")
    n, nch = 8192, 2
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = libxaac_amd.XaacContext(0, stream.cuda_stream)
    st0, _ = libxaac_amd.peak_limiter_init(nch, 48000)
    raw = np.frombuffer(bytes(st0), np.uint8)
    q = torch.full((n * nch,), 2, dtype=torch.int8, device=dev)
    pcm = torch.zeros(n * 1024 * nch, dtype=torch.int16, device=dev)
    ws = torch.zeros(ctx.peak_limiter_workspace_bytes(n), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for kind, std in (("quiet", 2.0 ** 24), ("mixed", 2.0 ** 27.5), ("loud", 2.0 ** 30)):
        if os.environ.get("XL_KIND", kind) != kind:   # one kind only (for rocprofv3 --pmc passes)
            continue
        state = torch.from_numpy(np.tile(raw, (n, 1))).to(dev)
        status = torch.zeros(n, dtype=torch.int32, device=dev)
        ts = []
        for it in range(6):
            x = (torch.randn(n * 1024 * nch, device=dev, generator=g) * std).clamp(-2.0 ** 31, 2.0 ** 31 - 256).to(torch.int32)
            torch.cuda.synchronize()
            ev[0].record(stream)
            ctx.peak_limiter_process_batch(x, q, state, nch, ws, pcm16=pcm, status=status)
            ev[1].record(stream)
            torch.cuda.synchronize()
            ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
        sv = state.cpu().numpy()
        mg = np.ascontiguousarray(sv[:, 24:28]).view(np.float32)
        print("%s %s: us per batch %s | min_gain mean %.3f" % (tag, kind, " ".join("%.0f" % t for t in ts), mg.mean()))
        if any("XL_PROFILE" in f for f in flags):   # the status array carries the phase timers (cycles summed over streams)
            cyc = status.cpu().numpy()[:16].view(np.uint64).astype(np.float64) / (6 * n)
            names = ["magnitudes + W", "sliding max", "tracking", "gain recursion", "walk (fallback)", "apply + state"]
            print("   cycles per stream-frame: " + ", ".join("%s %.0f" % (names[i], cyc[i]) for i in range(6)))


main()
