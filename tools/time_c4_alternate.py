#!/usr/bin/env python3
"""Developer tool: bench.py's C4 steps on one HIP stream against the same steps dealt out over S streams (step i on stream
i % S; the stream sets are independent batches, so neighbouring steps share nothing).  python tools/time_c4_alternate.py [S...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import libxaac_amd
import bench


def main():
    dev = torch.device("cuda:0")
    s0 = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(s0)
    ctx0 = libxaac_amd.XaacContext(0, s0.cuda_stream)
    sets = 4
    job = bench.Workload("c4", torch, libxaac_amd, ctx0, dev, s0, sets, 0)
    el, k = job.run(40, 4, torch.cuda.synchronize)
    print("one stream: ms/step %.4f (events %.4f)" % (el / 40 * 1e3, k))
    n = bench.FRAMES_PER_STEP
    for S in [int(a) for a in sys.argv[1:]] or [2]:
        assert sets % S == 0
        streams = [s0] + [torch.cuda.Stream(device=dev) for _ in range(S - 1)]
        ctxs = [ctx0] + [libxaac_amd.XaacContext(0, s.cuda_stream) for s in streams[1:]]
        wss = [job._workspace(n) for _ in range(S)]
        sts = [torch.zeros_like(job.status) for _ in range(S)]
        ists = [torch.zeros_like(job.imdct_status) for _ in range(S)]

        def step(i):
            q = i % S
            job.ctx = ctxs[q]
            with torch.cuda.stream(streams[q]):
                job.launch(job.batches[i % sets], i // sets, ws=wss[q], status=sts[q], imdct_status=ists[q])

        for i in range(8):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            step(8 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40 * 1e3
        print("%d streams: ms/step %.4f refused %s" % (S, dt, [int((s != 0).sum()) for s in sts]))
        job.ctx = ctx0


if __name__ == "__main__":
    main()
