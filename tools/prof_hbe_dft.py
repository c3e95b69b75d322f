#!/usr/bin/env python3
"""Stage timers of the DFT transposer's kernel (xaac_hbe_dft_core_kernel): builds the library with -DXE_PROFILE and prints thread
0's cycles per channel-frame in: front (loads + synthesis bank), twiddles, window + clear, forward transform, polar form,
stretch, inverse transform + overlap-add.  Developer tool; run on the GPU box."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
src = os.path.join(ROOT, "libxaac_amd", "csrc")
out = os.path.join(ROOT, "libxaac_amd", "libxaac_amd_prof.so")   # (built where hipcc is; travels to the GPU box with the snapshot)
files = [f for f in os.listdir(src) if f.endswith(".hip")] + ["xaac_abi.cpp"]
if "--build" in sys.argv or not os.path.exists(out):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-w", "-I",
                           os.path.join(src, "build"), "-DXE_PROFILE", "-shared", "-x", "hip"] + [os.path.join(src, f) for f in files] + ["-o", out])
    if "--build" in sys.argv:
        sys.exit(0)
import torch
import libxaac_amd
libxaac_amd.library_path = lambda: out
import test_hbe_dft as td
dev = torch.device("cuda:0")
ctx = libxaac_amd.XaacContext(0, None)
lib = ctypes.CDLL(out)
G = np.load(td.GOLDEN)
n = 4096
for case in [int(c) for c in G["cases"]]:
    st0, cfg, coef = td.golden_case(G, case)
    rng = np.random.default_rng(4000 + case)
    q, before, ovs, pitch = td.golden_inputs(case, 0, rng)
    tile = lambda a: torch.from_numpy(np.ascontiguousarray(np.stack([a] * n))).to(dev)
    st = tile(np.frombuffer(bytes(st0), np.uint8))
    cfg_tab = torch.from_numpy(np.frombuffer(bytes(cfg), np.uint8).copy()[None]).to(dev)
    cre, cim = torch.from_numpy(coef[0][None]).to(dev), torch.from_numpy(coef[1][None]).to(dev)
    qre, qim, pvr, pvi = tile(q[0]), tile(q[1]), tile(before[0]), tile(before[1])
    o = torch.full((n,), ovs, dtype=torch.int32, device=dev)
    pt = torch.full((n,), pitch, dtype=torch.int32, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    step = lambda: ctx.hbe_dft_apply_batch(qre, qim, cfg_tab, cre, cim, st, pvr, pvi, status, pitch_in_bins=pt, oversampling=o)
    step(); ctx.sync()
    lib.xaac_debug_hbe_prof(None, 1)
    steps = 3
    for _ in range(steps):
        step()
    ctx.sync()
    acc = (ctypes.c_ulonglong * 8)()
    lib.xaac_debug_hbe_prof(acc, 0)
    v = np.array(list(acc), np.float64) / (steps * n)
    names = ("twiddles", "window+clear", "fwd transform", "polar", "stretch", "inverse+ola", "front")
    print(td.CASES[case], " ".join("%s %.0f (%.0f%%)" % (nm, x, 100 * x / v[:7].sum()) for nm, x in zip(names, v[:7])), "total %.0f cycles" % v[:7].sum())
