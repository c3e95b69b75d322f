#!/usr/bin/env python3
"""Stage timers of the transposer's products kernel (xaac_hbe_post_kernel): builds the library with -DXE_PROFILE and prints
thread 0's cycles per channel-frame in (0) normalised-sample planes, (1) column blocks + cross products, (2) rows 0..31
(gather + output rotation), (3) rows 32..63.  Developer tool; run on the GPU box."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import libxaac_amd
src = os.path.join(ROOT, "libxaac_amd", "csrc")
out = "/tmp/libxaac_amd_prof.so"
files = [f for f in os.listdir(src) if f.endswith(".hip")] + ["xaac_abi.cpp"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-w",
                       "-DXE_PROFILE", "-shared", "-x", "hip"] + [os.path.join(src, f) for f in files] + ["-o", out])
libxaac_amd.library_path = lambda: out
from make_golden_hbe import state_from_params
dev = torch.device("cuda:0")
ctx = libxaac_amd.XaacContext(0, None)
lib = ctypes.CDLL(out)
n = 8192
rng = np.random.default_rng(0)
for label, par in (("size 12, x2 x3", [12, 6, 15, 41, 15, 29, 41, 0, 0, 0, 3]), ("size 8, x2 x3 x4", [8, 2, 9, 31, 9, 15, 27, 31, 0, 0, 4])):
    hst = torch.from_numpy(np.stack([np.frombuffer(bytes(state_from_params(par)), np.uint8)] * n)).to(dev)
    qre = torch.from_numpy((rng.standard_normal((n, 32, 64)) * 1000).astype(np.float32)).to(dev)
    qim = torch.from_numpy((rng.standard_normal((n, 32, 64)) * 1000).astype(np.float32)).to(dev)
    pvr, pvi = torch.zeros_like(qre), torch.zeros_like(qre)
    stat = torch.zeros(n, dtype=torch.int32, device=dev)
    for _ in range(2):
        ctx.hbe_apply_batch(qre, qim, hst, pvr, pvi, stat)
    ctx.sync()
    lib.xaac_debug_hbe_prof(None, 1)
    steps = 4
    for _ in range(steps):
        ctx.hbe_apply_batch(qre, qim, hst, pvr, pvi, stat)
    ctx.sync()
    acc = (ctypes.c_ulonglong * 8)()
    lib.xaac_debug_hbe_prof(acc, 0)
    v = np.array(list(acc), np.float64)[:4] / (steps * n)
    print(label, " ".join("%s %.0f (%.0f%%)" % (nm, x, 100 * x / v.sum()) for nm, x in zip(("planes", "blocks", "rows0-31", "rows32-63"), v)), "total %.0f cycles" % v.sum())
