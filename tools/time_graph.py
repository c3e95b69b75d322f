#!/usr/bin/env python3
"""Developer tool (GPU box): the C4 step's six launches replayed as one HIP graph against the same launches made directly,
on one HIP stream, for the small-batch sweep's sizes and the full batch ("the six launches as one HIP
graph for batches <= 1024").  A step's pointers differ per stream set and per side-info frame, so one graph per (set, frame
row) is captured; the timed loops walk them in the bench's order.  Prints ms per step for both ways."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import libxaac_amd
    import bench
    dev = torch.device("cuda:0")
    st, ctx = bench.extra_lanes(torch, libxaac_amd, dev, 1)[0]
    wl = bench.Workload("c4", torch, libxaac_amd, ctx, dev, st, 4, 1234, hip_streams=1)
    sets, nfr = len(wl.batches), len(wl.batches[0]["frames"])
    steps = int(os.environ.get("STEPS", "200"))
    out = {}
    for k in (256, 1024, 4096, None):
        ws = wl._workspace(k or bench.FRAMES_PER_STEP)

        def direct(i):
            wl.launch(wl.batches[i % sets], i // sets, k=k, ws=ws, ctx=ctx)

        with torch.cuda.stream(st):
            for i in range(2 * sets * nfr):
                direct(i)
        torch.cuda.synchronize()
        graphs = {}
        for i in range(sets * nfr):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                direct(i)
            graphs[(i % sets, (i // sets) % nfr)] = g
        torch.cuda.synchronize()

        def timed(fn):
            with torch.cuda.stream(st):
                for i in range(16):
                    fn(i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    fn(i)
                torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3

        ms_d = timed(direct)
        ms_g = timed(lambda i: graphs[(i % sets, (i // sets) % nfr)].replay())
        bad = float((wl.status != 0).float().mean().item())
        out[k or bench.FRAMES_PER_STEP] = {"direct_ms": round(ms_d, 4), "graph_ms": round(ms_g, 4), "refused": bad}
        print(k or bench.FRAMES_PER_STEP, out[k or bench.FRAMES_PER_STEP], flush=True)


if __name__ == "__main__":
    main()
