#!/usr/bin/env python3
"""Developer tool: time xaac_sbr_lp_process_batch for library variants built with extra -D flags."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def build(tag, flags):
    src = os.path.join(ROOT, "libxaac_amd", "csrc")
    out = os.path.join(ROOT, "libxaac_amd", "libxaac_amd_%s.so" % tag)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip"]
                          + flags + [os.path.join(src, f) for f in ("imdct_kernel.hip", "sbr_qmf_kernel.hip", "sbr_core_kernel.hip", "sbr_ps_kernel.hip", "limiter_kernel.hip", "xaac_abi.cpp")] + ["-o", out])
    return out

def main():
    tag, flags = sys.argv[1], sys.argv[2:]
    import torch
    import libxaac_amd
    out = build(tag, flags)
    libxaac_amd.library_path = lambda: out
    import bench
    dev = torch.device("cuda:0")
    n = bench.FRAMES_PER_STEP * bench.CH
    b = bench.make_inputs_c3(torch, dev, 1, 0)[0]
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = libxaac_amd.XaacContext(0, stream.cuda_stream)
    ws = torch.zeros(ctx.sbr_lp_workspace_bytes(n), dtype=torch.uint8, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for i in range(8):
        ctx.imdct_process_batch(b["spec"], b["ics"], b["overlap"], b["state"], None, b["core_pcm"], None, ch_fac=1, pcm_mode=libxaac_amd.PCM_SBR)
        ev[0].record(stream)
        ctx.sbr_lp_process_batch(b["core_pcm"], b["hdr"], b["frames"][i % 4], b["sbr_state"], b["pcm"], ws, None, in_ch_fac=1, out_ch_fac=2)
        ev[1].record(stream)
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    print(tag, flags, "sbr_lp_process_batch ms:", " ".join("%.3f" % t for t in ts))

main()
