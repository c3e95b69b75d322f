#!/usr/bin/env python3
"""Developer tool (GPU box): the PVC kernel against the oracle word by word on the fixture's chains; prints the frames that differ."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, libxaac_amd
import pvc_structs as ps
fn = ps.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "xo_pvc_process")
dev = torch.device("cuda:0"); ctx = libxaac_amd.XaacContext(0, None)
seeds, frames = list(range(5000, 5048)), 24
chains = [ps.chain(s, frames) for s in seeds]
n = len(chains)
state = torch.zeros((n, libxaac_amd.PVC_STATE_BYTES), dtype=torch.uint8, device=dev)
sts = [ps.PvcState() for _ in range(n)]
for fr in range(frames):
    items = [c[fr] for c in chains]
    frame = torch.from_numpy(np.stack([np.frombuffer(bytes(i[0]), np.uint8) for i in items])).to(dev)
    re = torch.from_numpy(np.stack([i[1] for i in items])).to(dev); im = torch.from_numpy(np.stack([i[2] for i in items])).to(dev)
    clear = [k for k, i in enumerate(items) if i[3]]
    if clear: state[clear, ps.PvcState.prev_pvc_flg.offset] = 0
    out = torch.zeros((n, 16, 64), dtype=torch.float32, device=dev)
    ctx.pvc_process_batch(frame, re, im, state, out, None); ctx.sync()
    o = out.cpu().numpy()
    for k in range(n):
        f, r, i_, cl = items[k]
        if cl: sts[k].prev_pvc_flg = 0
        want = np.zeros((16, 64), np.float32)
        fn(ctypes.byref(f), r.ctypes.data_as(ps.PF), i_.ctypes.data_as(ps.PF), ctypes.byref(sts[k]), want.ctypes.data_as(ps.PF))
        bad = np.argwhere(want.view(np.uint32) != o[k].view(np.uint32))
        if bad.size:
            print("chain", k, "frame", fr, "mode", f.pvc_mode, "ns", f.ns_mode, "rate", f.pvc_rate, "lp", f.low_power, "first", f.first_bnd_idx, "slot0", f.first_pvc_timeslot,
                  "n_bad", len(bad), "at", bad[:3].tolist(), want[tuple(bad[0])], o[k][tuple(bad[0])], "ids", list(f.pvc_id)[:4])
