#!/usr/bin/env python3
"""Developer tool: the C4 chain of bench.py as one batch on one HIP stream against P parts on P streams
(python tools/time_c4_streams.py [parts...]); run on the GPU box."""
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
    job = bench.Workload("c4", torch, libxaac_amd, ctx0, dev, s0, 2, 0)
    el, k = job.run(30, 4, torch.cuda.synchronize)
    print("one stream: ms/step %.4f" % k)
    n = bench.FRAMES_PER_STEP
    x = libxaac_amd
    for parts in [int(a) for a in sys.argv[1:]] or [2]:
        streams = [s0] + [torch.cuda.Stream(device=dev) for _ in range(parts - 1)]
        ctxs = [ctx0] + [libxaac_amd.XaacContext(0, s.cuda_stream) for s in streams[1:]]
        per = n // parts
        wss = [torch.zeros(ctx0.sbr_hq_workspace_bytes(per, True), dtype=torch.uint8, device=dev) for _ in range(parts)]

        def step(i):
            b = job.batches[i % len(job.batches)]
            fr, pfr = b["frames"][(i // len(job.batches)) % len(b["frames"])]
            start = torch.cuda.Event()
            start.record(s0)
            for p in range(parts):
                sl = lambda t, w: t.view(n, -1)[p * per:(p + 1) * per].reshape(-1) if t.dim() == 1 else t[p * per:(p + 1) * per]
                c = ctxs[p]
                if p:
                    streams[p].wait_event(start)
                with torch.cuda.stream(streams[p]):
                    c.imdct_process_batch(sl(b["spec"], 1), sl(b["ics"], 1), sl(b["overlap"], 1), sl(b["state"], 1), None,
                                          sl(b["core_pcm"], 1024), None, ch_fac=1, pcm_mode=x.PCM_SBR,
                                          status=sl(job.imdct_status, 1))
                    c.sbr_hq_process_batch(sl(b["core_pcm"], 1024), sl(b["hdr"], 1), sl(fr, 1), sl(b["sbr_state"], 1),
                                           sl(b["pcm"], 4096), wss[p], sl(pfr, 1), sl(b["ps_state"], 1), sl(job.status, 1))
                if p:
                    e = torch.cuda.Event()
                    e.record(streams[p])
                    s0.wait_event(e)

        for i in range(4):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(30):
            step(4 + i)
        torch.cuda.synchronize()
        print("%d streams: ms/step %.4f refused %s" % (parts, (time.perf_counter() - t0) / 30 * 1e3, int((job.status != 0).sum())))


if __name__ == "__main__":
    main()
