#!/usr/bin/env python3
"""Stage timers of the Path A core kernel: builds the product library with -DXE_PROFILE (lane-0 cycle counts per stage,
summed into the status buffer), runs the eSBR chain of tools/bench_new_kernels.py's inputs and prints cycles per
channel-frame.  Developer tool; run on the GPU box."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
NAMES = ["side info to LDS", "history clear + sbr_qmf_out init", "generate_hf", "env_calc: rest (reset, harmonics map, tail)", "regroup (+PS rows)",
         "history shift", "env: band map (lane 0)", "env: energies", "env: gains", "env: limiter", "env: apply + sinusoids"]


PS_NAMES = ["state clear above the SBR range", "hybrid analysis + history", "band powers", "transient detector",
            "decorrelator, hybrid sub-bands", "decorrelator, QMF bands", "rotation (all envelopes)", "hybrid synthesis"]


def main_ps(torch, libxaac_amd, ctx, dev, n, rng, c, make_side, core, pcm, ws, status):
    """python tools/prof_esbr_core.py ps: the float parametric-stereo kernel's stages on HE-AACv2 side info"""
    from esbr_structs import new_state, new_ps_state, EsbrSide
    from test_esbr_ps_oracle_vs_reference import fuzz_ps_frame
    recs = [r for r in c.read_records(os.path.join(ROOT, "tests", "golden", "sbr_hq_ps_records.bin.gz")) if r["ps"] and r["frame"].apply_processing][:64]
    hs, fs, sds, pfs = [], [], [], []
    for r in recs:
        h, f = c.Header.from_buffer_copy(bytes(r["header"])), c.Frame.from_buffer_copy(bytes(r["frame"]))
        sd = make_side(rng, h, f, [0] * 10, 0, 0, False)
        sd.reset_flag = 1
        pf = fuzz_ps_frame(rng, c.PsFrame.from_buffer_copy(bytes(r["ps_frame"])), 0)
        for x, lst in ((h, hs), (f, fs), (sd, sds), (pf, pfs)):
            lst.append(np.frombuffer(bytes(x), np.uint8))
    tile = lambda xs: torch.from_numpy(np.stack([xs[i % len(xs)] for i in range(n)])).to(dev)
    hd, fr, sd, pf = tile(hs), tile(fs), tile(sds), tile(pfs)
    st = torch.from_numpy(np.stack([np.frombuffer(bytes(new_state()), np.uint8)] * n)).to(dev)
    pst = torch.from_numpy(np.stack([np.frombuffer(bytes(new_ps_state()), np.uint8)] * n)).to(dev)
    pcm_r = torch.zeros_like(pcm)
    ctx.esbr_sbr_process_batch(core, hd, fr, sd, st, pcm, ws, status, pf, pst, pcm_r)
    ctx.sync()
    status.zero_()
    off = EsbrSide.reset_flag.offset
    sd.view(n, -1)[:, off:off + 2] = 0
    steps = 4
    for _ in range(steps):
        ctx.esbr_sbr_process_batch(core, hd, fr, sd, st, pcm, ws, status, pf, pst, pcm_r)
    ctx.sync()
    acc = status.cpu().numpy()[:64].view(np.uint64).astype(np.float64)[16:24] / (steps * n)
    for nm, v in zip(PS_NAMES, acc):
        print("%-36s %9.0f cycles/stream-frame %5.1f%%" % (nm, v, 100 * v / acc.sum()))
    print("total %.0f cycles" % acc.sum())


def main():
    import torch
    import libxaac_amd
    src = os.path.join(ROOT, "libxaac_amd", "csrc")
    out = os.path.join(ROOT, "libxaac_amd", "libxaac_amd_prof.so")
    files = [f for f in os.listdir(src) if f.endswith(".hip")] + ["xaac_abi.cpp"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-DXE_PROFILE", "-shared", "-x", "hip"] + [os.path.join(src, f) for f in files] + ["-o", out])
    libxaac_amd.library_path = lambda: out
    import sbr_capture as c
    from esbr_structs import new_state
    from test_esbr_core_oracle_vs_reference import make_side
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, None)
    n = 8192
    rng = np.random.default_rng(0)
    recs = [r for r in c.read_records(os.path.join(ROOT, "tests", "golden", "sbr_lp_records.bin.gz")) if r["frame"].apply_processing][:64]
    hs, fs, sds = [], [], []
    for r in recs:
        h, f = c.Header.from_buffer_copy(bytes(r["header"])), c.Frame.from_buffer_copy(bytes(r["frame"]))
        sd = make_side(rng, h, f, [0] * 10, 0, 0, False)
        sd.reset_flag = 1
        hs.append(np.frombuffer(bytes(h), np.uint8)), fs.append(np.frombuffer(bytes(f), np.uint8)), sds.append(np.frombuffer(bytes(sd), np.uint8))
    tile = lambda xs: torch.from_numpy(np.stack([xs[i % len(xs)] for i in range(n)])).to(dev)
    hd, fr, sd = tile(hs), tile(fs), tile(sds)
    st = torch.from_numpy(np.stack([np.frombuffer(bytes(new_state()), np.uint8)] * n)).to(dev)
    core = torch.from_numpy((rng.uniform(-1, 1, (n, 1024)) * 20000).astype(np.float32)).to(dev)
    pcm = torch.zeros((n, 2048), dtype=torch.float32, device=dev)
    ws = torch.zeros(ctx.esbr_workspace_bytes(n), dtype=torch.uint8, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    if len(sys.argv) > 1 and sys.argv[1] == "ps":
        return main_ps(torch, libxaac_amd, ctx, dev, n, rng, c, make_side, core, pcm, ws, status)
    import esbr_structs
    ctx.esbr_sbr_process_batch(core, hd, fr, sd, st, pcm, ws, status)   # first frame: reset flag set, limiter tables built
    ctx.sync()
    status.zero_()
    off = esbr_structs.EsbrSide.reset_flag.offset
    sd.view(n, -1)[:, off:off + 2] = 0
    steps = 4
    for _ in range(steps):
        ctx.esbr_sbr_process_batch(core, hd, fr, sd, st, pcm, ws, status)
    ctx.sync()
    acc = status.cpu().numpy()[:32].view(np.uint64).astype(np.float64)[:11] / (steps * n)
    for nm, v in zip(NAMES, acc):
        print("%-36s %9.0f cycles/channel-frame %5.1f%%" % (nm, v, 100 * v / acc.sum()))
    print("total %.0f cycles" % acc.sum())


if __name__ == "__main__":
    main()
