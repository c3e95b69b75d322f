#!/usr/bin/env python3
"""bench.py -- decoded audio frames/s of the MI355X transform back-end.

Workload (BASELINE.json configs[1], SURVEY.md §8d "C2"): AAC-LC 48 kHz stereo,
batch = 8192 frames per step = 16384 channel-frames through the 1024-sample
IMDCT + window/overlap-add kernel with the fused PCM16 hand-off, stereo
interleaved output.  Synthetic spectra (seeded, uniform in +-2^17 below bin 640,
zero above; ONLY_LONG, window shape alternating per frame), overlap state
carried in HBM.  Each rank owns `--sets` x 8192 independent streams and decodes
one frame of 8192 of them per step, round-robin, so the working set (> 256 MiB)
streams from HBM rather than from the Infinity Cache.

A "step" = one pass of the hot path over one batch (one kernel launch).
Inputs are resident in HBM before the timed region.  `value` = frames decoded by
all ranks / wall time (max over ranks).  `roofline.achieved` = algorithmic bytes
per launch (20480 B per stereo frame) / mean kernel duration from
HIP events recorded on the launch stream.  `cpu_baseline` = the same workload on
the host cores (the compiled reference when oracle/_ref travelled with the repo,
else the bit-exact C restatement), bounded sample, rank 0 at N=1 only.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAMES_PER_STEP = 8192          # stereo frames per batch (BASELINE configs[1])
CH = 2
ALG_BYTES_PER_FRAME = 20480     # R spec 2x4096 + ovl 2x2048, W pcm16 2x2048 + ovl 2x2048
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_hbm_latest.json")   # written from the rocprofv3 --pmc passes


def measured_traffic():
    """HBM bytes per launch from the last committed PMC profile of this kernel (rocprofv3
    FETCH_SIZE x2 + WRITE_SIZE, calibrated as MI355X_MICROARCH.md prescribes), or None."""
    try:
        with open(PMC_FILE) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


# ---- workload C2L: C2 + the AAC-LC post stage (peak limiter + PCM16), SURVEY.md §8 row f2 ---------------------
# per stereo frame: IMDCT R spec + ovl, W out32 + ovl; limiter R out32 + delay line + window, W out32 + PCM16 + both
C2L_ALG_BYTES_PER_FRAME = (8192 + 4096) + (8192 + 4096) + (8192 + 1920 + 960) + (8192 + 4096 + 1920 + 960)
C2L_LOUD_EVERY = 4              # every 4th stream is a hot master: its spectra decode to peaks above full scale


def make_inputs(torch, device, sets, seed, limiter=False):
    """`sets` batches of 16384 channel-frames, generated on the device."""
    g = torch.Generator(device=device)
    g.manual_seed(0xC0FFEE + seed)
    n = FRAMES_PER_STEP * CH
    batches = []
    for s in range(sets):
        spec = torch.randint(-(1 << 17), 1 << 17, (n, 1024), generator=g, device=device, dtype=torch.int32)
        if limiter:   # x16 on the hot streams: ~12 % of their samples leave the IMDCT above 2^31 (tools/amp_probe.py)
            hot = (torch.arange(n, device=device) // CH) % C2L_LOUD_EVERY == C2L_LOUD_EVERY - 1
            spec[hot] *= 16
        spec[:, 640:] = 0
        ics = torch.zeros((n, 2), dtype=torch.uint8, device=device)
        batches.append({
            "spec": spec, "ics": ics,
            "overlap": torch.zeros((n, 512), dtype=torch.int32, device=device),
            "state": torch.zeros((n, 2), dtype=torch.uint8, device=device),
            "pcm": torch.zeros(n * 1024, dtype=torch.int16, device=device),
        })
        if limiter:
            import libxaac_amd
            st0, _ = libxaac_amd.peak_limiter_init(CH, 48000)
            raw = np.frombuffer(bytes(st0), np.uint8)
            batches[-1].update({
                "out32": torch.zeros(n * 1024, dtype=torch.int32, device=device),
                "qshift": torch.zeros(n, dtype=torch.int8, device=device),
                "lim_state": torch.from_numpy(np.tile(raw, (FRAMES_PER_STEP, 1))).to(device)})
    return batches


# ---- workload C3: HE-AACv1 stereo (BASELINE.json configs[2]) ------------------------------------------------
C3_ALG_BYTES_PER_CH = (4096 + 2048 + 336 + 1072 + 7300) + (2048 + 7300 + 4096)   # R + W per channel-frame, DESIGN.md


def make_inputs_c3(torch, device, sets, seed):
    """Core spectra as C2 (24 kHz core: bins < 512 populated) + SBR side info cycled from the committed
    reference-captured frames (tests/golden/sbr_lp_records.bin.gz), grouped by stream configuration."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sbr_capture as cap
    recs = [r for r in cap.read_records(os.path.join(ROOT, "tests", "golden", "sbr_lp_records.bin.gz")) if r["enh"] == 0]
    groups = {}
    for r in recs:
        groups.setdefault(bytes(r["header"]), []).append(r)
    groups = list(groups.values())
    n = FRAMES_PER_STEP * CH
    g = torch.Generator(device=device)
    g.manual_seed(0xC0FFEE + seed)
    hdr = np.zeros((n, 336), np.uint8); st0 = np.zeros((n, 7300), np.uint8)
    frames = []
    for c in range(n):
        grp = groups[(c // CH) % len(groups)]
        hdr[c] = np.frombuffer(bytes(grp[0]["header"]), np.uint8)
        st0[c] = np.frombuffer(bytes(grp[0]["st0"]), np.uint8)      # the reference's own initial state
    nvar = 4
    for v in range(nvar):
        fr = np.zeros((n, 1072), np.uint8)
        for c in range(n):
            grp = groups[(c // CH) % len(groups)]
            fr[c] = np.frombuffer(bytes(grp[(c + v) % len(grp)]["frame"]), np.uint8)
        frames.append(torch.from_numpy(fr).to(device))
    batches = []
    for s in range(sets):
        spec = torch.randint(-(1 << 17), 1 << 17, (n, 1024), generator=g, device=device, dtype=torch.int32)
        spec[:, 512:] = 0
        batches.append({"spec": spec, "ics": torch.zeros((n, 2), dtype=torch.uint8, device=device),
                        "overlap": torch.zeros((n, 512), dtype=torch.int32, device=device),
                        "state": torch.zeros((n, 2), dtype=torch.uint8, device=device),
                        "core_pcm": torch.zeros(n * 1024, dtype=torch.int16, device=device),
                        "hdr": torch.from_numpy(hdr).to(device), "frames": frames,
                        "sbr_state": torch.from_numpy(st0).to(device),
                        "pcm": torch.zeros(n * 2048, dtype=torch.int16, device=device)})
    return batches


# ---- workload C4: HE-AACv2 mono + parametric stereo (BASELINE.json configs[3]) -----------------------------
# R: spec 4096 + overlap 2048 + SBR header 336 / frame 1072 / state 7300 + PS frame 972 / state 7764;
# W: overlap 2048 + SBR state 7300 + PS state 7764 + PCM16 L,R 8192 -- per stream and frame, DESIGN.md
C4_ALG_BYTES_PER_STREAM = (4096 + 2048 + 336 + 1072 + 7300 + 972 + 7764) + (2048 + 7300 + 7764 + 8192)


def make_inputs_c4(torch, device, sets, seed):
    """One mono core channel per stream (24 kHz core: bins < 512 populated) + SBR and PS side info cycled from
    the committed reference-captured HE-AACv2 frames (tests/golden/sbr_hq_ps_records.bin.gz)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sbr_capture as cap
    recs = [r for r in cap.read_records(os.path.join(ROOT, "tests", "golden", "sbr_hq_ps_records.bin.gz")) if r["enh"] == 0]
    groups = {}
    for r in recs:
        groups.setdefault(bytes(r["header"]), []).append(r)
    groups = list(groups.values())
    n = FRAMES_PER_STEP
    g = torch.Generator(device=device)
    g.manual_seed(0xC0FFEE + seed)
    rows = lambda key, size, pick: torch.from_numpy(np.stack(
        [np.frombuffer(bytes(pick(groups[c % len(groups)], c)[key]), np.uint8) for c in range(n)])).to(device)
    hdr = rows("header", 336, lambda grp, c: grp[0])
    st0 = rows("st0", 7300, lambda grp, c: grp[0])
    ps0 = rows("ps0", 7764, lambda grp, c: grp[0])
    frames = [(rows("frame", 1072, lambda grp, c, v=v: grp[(c + v) % len(grp)]),
               rows("ps_frame", 972, lambda grp, c, v=v: grp[(c + v) % len(grp)])) for v in range(4)]
    batches = []
    for s in range(sets):
        spec = torch.randint(-(1 << 17), 1 << 17, (n, 1024), generator=g, device=device, dtype=torch.int32)
        spec[:, 512:] = 0
        batches.append({"spec": spec, "ics": torch.zeros((n, 2), dtype=torch.uint8, device=device),
                        "overlap": torch.zeros((n, 512), dtype=torch.int32, device=device),
                        "state": torch.zeros((n, 2), dtype=torch.uint8, device=device),
                        "core_pcm": torch.zeros(n * 1024, dtype=torch.int16, device=device),
                        "hdr": hdr, "frames": frames, "sbr_state": st0.clone(), "ps_state": ps0.clone(),
                        "pcm": torch.zeros(n * 4096, dtype=torch.int16, device=device)})
    return batches


def cpu_baseline_sbr(workload, seconds_budget=10.0):
    """CPU baseline of the SBR workloads: the compiled reference's own ixheaacd_sbr_dec driven through
    oracle/ref_sbr_adapter.c (kind "reference") or, where oracle/_ref did not travel, the bit-exact restatement
    (kind "port"), on the committed reference-captured frames: every host thread runs a C loop over its own shard
    of channel-frames (arrays of the boundary structs, state reset per pass)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import sbr_capture as cap
    hq = workload == "c4"
    recs = [r for r in cap.read_records(os.path.join(ROOT, "tests", "golden",
            "sbr_hq_ps_records.bin.gz" if hq else "sbr_lp_records.bin.gz")) if r["enh"] == 0]
    ref = oracle_lib.load_reference()
    name = "sbr_dec_hq_batch" if hq else "sbr_dec_lp_batch"
    if ref is not None and hasattr(ref.lib, "ref_" + name):
        kind, fn = "reference", getattr(ref.lib, "ref_" + name)
    else:
        kind, fn = "port", getattr(oracle_lib.load_oracle().lib, "xo_" + name)
    fn.restype = ctypes.c_int
    cores = os.cpu_count() or 1
    per_thread = 512                   # channel-frames per thread per pass (thread start-up amortised)
    pick = [recs[i % len(recs)] for i in range(per_thread)]
    blob = lambda key: b"".join(bytes(r[key]) for r in pick)
    hdr, frm, st0 = blob("header"), blob("frame"), blob("st0")
    pin = np.concatenate([r["pcm_in"] for r in pick]).astype(np.int16)
    psf, ps0 = (blob("ps_frame"), blob("ps0")) if hq else (None, None)

    def shard(_):
        st = ctypes.create_string_buffer(st0, len(st0))
        out = np.zeros(per_thread * (4096 if hq else 2048), np.int16)
        if hq:
            ps = ctypes.create_string_buffer(ps0, len(ps0))
            fn(per_thread, hdr, frm, st, psf, ps, pin.ctypes.data, out.ctypes.data)
        else:
            fn(per_thread, hdr, frm, st, pin.ctypes.data, out.ctypes.data)

    fn.argtypes = ([ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p,
                    ctypes.c_void_p, ctypes.c_void_p] if hq else
                   [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p])

    def one_pass(nthreads):
        t0 = time.perf_counter()
        th = [threading.Thread(target=shard, args=(t,)) for t in range(nthreads)]
        [t.start() for t in th]
        [t.join() for t in th]
        return time.perf_counter() - t0

    one_pass(cores)
    t1 = min(one_pass(1) for _ in range(2))
    best, spent, passes = 1e9, 0.0, 0
    while spent < seconds_budget and passes < 400:
        dt = one_pass(cores)
        best, spent, passes = min(best, dt), spent + dt, passes + 1
    per_frame = 1.0 if hq else 0.5     # C3 counts stereo frames: two channel calls each
    return {"value": round(cores * per_thread * per_frame / best, 1), "unit": "frames/s", "cores": cores, "kind": kind,
            "value_1core": round(per_thread * per_frame / t1, 1),
            "sample": "%d ixheaacd_sbr_dec calls per thread per pass (C loop) on the committed reference-captured "
                      "frames, %d passes; SBR chain only, the core IMDCT is not in it" % (per_thread, passes)}


def cpu_baseline(seconds_budget=12.0, limiter=False):
    """Time the CPU path on a bounded sample of the same workload (all host cores,
    one contiguous shard of channel-frames per thread).  limiter: each thread also runs the reference's
    ixheaacd_peak_limiter_process + round16 over its shard's frames (workload C2L)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    P32, P16, PU8 = oracle_lib.P32, oracle_lib.P16, oracle_lib.PU8
    ref = oracle_lib.load_reference()
    kind = "reference" if ref is not None and hasattr(ref.lib, "ref_imdct_batch") else "port"
    orc = oracle_lib.load_oracle()
    cores = os.cpu_count() or 1
    per_thread = max(256, min(2048, 131072 // cores))   # channel-frames per thread per pass
    rng = np.random.default_rng(0xC0FFEE)
    n = per_thread * cores
    spec0 = rng.integers(-(1 << 17), 1 << 17, (n, 1024)).astype(np.int32)
    spec0[:, 640:] = 0
    ovl = np.zeros((n, 512), np.int32)
    pseq = np.zeros(n, np.int16); pshape = np.zeros(n, np.int16)
    seq = np.zeros(n, np.uint8); shape = (np.arange(n) % 2).astype(np.uint8)
    pcm = np.zeros((n, 1024), np.int16)
    spec = spec0.copy()

    if limiter:   # the limiter's cost does not depend on the signal (it is a per-sample loop either way)
        import limiter_cases as lc
        import libxaac_amd
        lim_batch = lc.bind(ref.lib if kind == "reference" else orc.lib, "ref" if kind == "reference" else "xo")[2]
        lim_state = (lc.LimiterState * (n // CH))()
        st0, _ = libxaac_amd.peak_limiter_init(CH, 48000)
        for i in range(n // CH):
            ctypes.memmove(ctypes.byref(lim_state[i]), ctypes.byref(st0), ctypes.sizeof(st0))
        lim_x = (rng.standard_normal(n * 1024) * 2.0 ** 27.5).clip(-2.0 ** 31, 2.0 ** 31 - 256).astype(np.int32)
        lim_q = np.full(n, 2, np.int8)
        lim_pcm = np.zeros(n * 1024, np.int16)

    def shard(t):
        a, b = t * per_thread, (t + 1) * per_thread
        p = oracle_lib._p
        if kind == "reference":
            ref.lib.ref_imdct_batch(per_thread, p(spec[a:b], P32), p(ovl[a:b], P32), p(pseq[a:b], P16),
                                    p(pshape[a:b], P16), p(seq[a:b], PU8), p(shape[a:b], PU8), p(pcm[a:b], P16))
        else:
            orc.lib.xo_imdct_batch(per_thread, p(spec[a:b], P32), p(ovl[a:b], P32), p(pseq[a:b], P16),
                                   p(pshape[a:b], P16), p(seq[a:b], PU8), p(shape[a:b], PU8), None,
                                   p(pcm[a:b], P16), None, 0)
        if limiter:
            lim_batch(per_thread // CH, 1024, CH, p(lim_x[a * 1024:b * 1024], P32), 1024 * CH, p(lim_q[a:b], oracle_lib.P8),
                      ctypes.cast(ctypes.byref(lim_state, (a // CH) * ctypes.sizeof(lc.LimiterState)),
                                  ctypes.POINTER(lc.LimiterState)), p(lim_pcm[a * 1024:b * 1024], P16))

    if kind == "reference":
        ref.lib.ref_imdct_batch.restype = None
        ref.lib.ref_imdct_batch.argtypes = [ctypes.c_int, P32, P32, P16, P16, PU8, PU8, P16]

    def one_pass(nthreads):
        if kind == "reference":
            spec[:] = spec0           # the reference transforms its input in place
        t0 = time.perf_counter()
        th = [threading.Thread(target=shard, args=(t,)) for t in range(nthreads)]
        [t.start() for t in th]
        [t.join() for t in th]
        return time.perf_counter() - t0

    one_pass(cores)                   # warm
    t1 = min(one_pass(1) for _ in range(2))
    passes, spent, best = 0, 0.0, 1e9
    while spent < seconds_budget and passes < 2000:
        dt = one_pass(cores)
        best = min(best, dt)
        spent += dt
        passes += 1
    frames = n / CH
    return {"value": round(frames / best, 1), "unit": "frames/s", "cores": cores, "kind": kind,
            "value_1core": round(per_thread / CH / t1, 1),
            "sample": "%d stereo frames (%d channel-frames) per pass, %d passes, one %d-channel-frame shard per "
                      "thread, same synthetic C2 input%s" % (frames, n, passes, per_thread,
                                                             " + limiter/round16 per frame" if limiter else "")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--sets", type=int, default=4, help="independent 8192-stream batches cycled per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["c2", "c2l", "c3", "c4"], default="c2",
                    help="c2: AAC-LC IMDCT+OLA (BASELINE configs[1], default); c2l: c2 + peak limiter + PCM16 (the "
                         "AAC-LC post stage); c3: HE-AACv1 stereo, IMDCT + LP-SBR; c4: HE-AACv2, mono IMDCT + HQ-SBR + "
                         "parametric stereo")
    args = ap.parse_args()

    import torch
    import libxaac_amd

    from libxaac_amd import dist as xdist
    rank, local_rank, world = xdist.env_rank()
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    assert torch.cuda.is_available(), "bench.py needs a GPU; the product has no CPU path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = xdist.init("nccl")       # RCCL; None at world 1

    stream = torch.cuda.Stream(device=dev)      # kernels AND timing events go on this one stream
    torch.cuda.set_stream(stream)
    ctx = libxaac_amd.XaacContext(local_rank, stream.cuda_stream)
    c3, c4, c2l = args.workload == "c3", args.workload == "c4", args.workload == "c2l"
    batches = (make_inputs_c3(torch, dev, args.sets, rank) if c3 else make_inputs_c4(torch, dev, args.sets, rank) if c4
               else make_inputs(torch, dev, args.sets, rank, limiter=c2l))
    for b in batches:                 # window shape alternates per frame; state follows
        b["ics"][:, 1] = (torch.arange(b["ics"].shape[0], device=dev) // (1 if c4 else CH) % 2).to(torch.uint8)
    ws = None
    if c3:
        ws = torch.zeros(ctx.sbr_lp_workspace_bytes(FRAMES_PER_STEP * CH), dtype=torch.uint8, device=dev)
    if c4:
        ws = torch.zeros(ctx.sbr_hq_workspace_bytes(FRAMES_PER_STEP, True), dtype=torch.uint8, device=dev)
    if c2l:
        ws = torch.zeros(ctx.peak_limiter_workspace_bytes(FRAMES_PER_STEP), dtype=torch.uint8, device=dev)

    def step(i, ev=None):
        b = batches[i % len(batches)]
        if ev is not None:
            ev[0].record(stream)
        if c3:
            # core decoder back-end: planar PCM16 with the SBR hand-off rounding, then the low-power SBR chain
            ctx.imdct_process_batch(b["spec"], b["ics"], b["overlap"], b["state"], None, b["core_pcm"], None,
                                    ch_fac=1, pcm_mode=libxaac_amd.PCM_SBR)
            ctx.sbr_lp_process_batch(b["core_pcm"], b["hdr"], b["frames"][(i // len(batches)) % len(b["frames"])],
                                     b["sbr_state"], b["pcm"], ws, None, in_ch_fac=1, out_ch_fac=CH)
        elif c4:
            # mono core back-end, then HQ SBR + parametric stereo -> L,R pairs
            ctx.imdct_process_batch(b["spec"], b["ics"], b["overlap"], b["state"], None, b["core_pcm"], None,
                                    ch_fac=1, pcm_mode=libxaac_amd.PCM_SBR)
            fr, pfr = b["frames"][(i // len(batches)) % len(b["frames"])]
            ctx.sbr_hq_process_batch(b["core_pcm"], b["hdr"], fr, b["sbr_state"], b["pcm"], ws, pfr, b["ps_state"])
        elif c2l:
            # AAC-LC tail as api.c runs it: WORD32 block + qshift_adj -> limiter in place -> interleaved PCM16.  The
            # block stays planar between the two (16-byte stores from the IMDCT; the limiter interleaves on the way out)
            ctx.imdct_process_batch(b["spec"], b["ics"], b["overlap"], b["state"], b["out32"], None, b["qshift"],
                                    ch_fac=1)
            ctx.peak_limiter_process_batch(b["out32"], b["qshift"], b["lim_state"], CH, ws, pcm16=b["pcm"], planar=True)
        else:
            ctx.imdct_process_batch(b["spec"], b["ics"], b["overlap"], b["state"], None, b["pcm"], None,
                                    ch_fac=CH, pcm_mode=libxaac_amd.PCM_LC)
        if ev is not None:
            ev[1].record(stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, events[i])
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = xdist.max_over_ranks(dist, elapsed, dev)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in events])) if events else float("nan")

    # the run stays honest: decode one more frame of set 0 and compare a slice with the oracle
    checked = None
    if rank == 0 and c3:
        checked = "see tests/test_sbr_gpu.py (bit-exact vs reference records and oracle chains)"
    if rank == 0 and c4:
        checked = "see tests/test_sbr_hq_gpu.py (bit-exact vs reference records and oracle chains)"
    if rank == 0 and c2l:
        st = batches[0]["lim_state"].cpu().numpy()
        mg = np.ascontiguousarray(st[:, 24:28]).view(np.float32).reshape(-1)
        checked = ("see tests/test_limiter_gpu.py (bit-exact vs reference vectors and oracle chains); streams limiting "
                   "in the last frame: %.1f %%" % (100.0 * float((mg < 1.0).mean())))
    if rank == 0 and not (c3 or c4 or c2l):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            orc = oracle_lib.load_oracle()
            b = batches[0]
            k = 256
            h = {n_: b[n_][:k].cpu().numpy() for n_ in ("spec", "ics", "overlap", "state")}
            want = orc.imdct_batch(h["spec"], h["ics"], h["overlap"], h["state"], ch_fac=CH)
            step(0)
            torch.cuda.synchronize()
            checked = bool(np.array_equal(b["pcm"][:k * 1024].cpu().numpy().reshape(k, 1024), want["pcm16"]) and
                           np.array_equal(b["overlap"][:k].cpu().numpy(), want["overlap"]))
        except Exception as e:  # the checker is optional for the measurement itself
            checked = "unavailable: %s" % e

    if rank == 0:
        frames = FRAMES_PER_STEP * args.steps * world
        alg_bytes = (C3_ALG_BYTES_PER_CH * CH if c3 else C4_ALG_BYTES_PER_STREAM if c4 else
                     C2L_ALG_BYTES_PER_FRAME if c2l else ALG_BYTES_PER_FRAME) * FRAMES_PER_STEP
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": ("decoded audio frames/s (1024-spl IMDCT + 32/64-band QMF + low-power SBR, HE-AACv1 stereo)" if c3
                       else "decoded audio frames/s (1024-spl IMDCT + complex QMF + HQ SBR + parametric stereo, "
                            "HE-AACv2)" if c4
                       else "decoded audio frames/s (1024-spl IMDCT+overlap-add + peak limiter + PCM16, AAC-LC stereo)"
                       if c2l else "decoded audio frames/s (1024-spl IMDCT+overlap-add, AAC-LC stereo)"),
            "value": round(frames / elapsed, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32 (+ f32/f64 gain smoothing)" if c2l else "int32",
            "data": "synthetic",
            "config": {"workload": ("C3: HE-AACv1 48 kHz stereo, batch=8192 frames/step: IMDCT+OLA -> QMF-32 analysis -> "
                                    "LPP HF generation + envelope adjustment (side info cycled from reference-captured "
                                    "frames) -> QMF-64 synthesis, %d stream sets cycled" % args.sets) if c3 else
                                   ("C4: HE-AACv2 48 kHz, batch=8192 frames/step (one mono core channel each): IMDCT+OLA "
                                    "-> complex QMF-32 analysis -> LPP transposer + envelope adjustment -> parametric "
                                    "stereo -> two complex QMF-64 synthesis banks (SBR / PS side info cycled from "
                                    "reference-captured frames), %d stream sets cycled" % args.sets) if c4 else
                                   ("C2L: C2 + the AAC-LC post stage, batch=8192 frames/step: IMDCT + window/overlap-add "
                                    "(WORD32 block + qshift_adj) -> peak limiter in place (5 ms look-ahead, attack / "
                                    "release smoothing) -> PCM16; every %dth stream decodes above full scale, %d stream "
                                    "sets cycled" % (C2L_LOUD_EVERY, args.sets)) if c2l else
                                   ("C2: AAC-LC 48 kHz stereo, batch=8192 frames/step (16384 channel-frames), "
                                    "ONLY_LONG 1024-pt IMDCT + window/overlap-add + PCM16, %d stream sets cycled"
                                    % args.sets),
                       "frames_per_step": FRAMES_PER_STEP, "channels": 1 if c4 else CH, "launch": ctx.last_launch(),
                       "sharding": "streams split across ranks, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": None if (c3 or c4 or c2l) else (measured_traffic() or {}).get("bytes_per_launch"),
                         "traffic_source": None if (c3 or c4 or c2l) else (measured_traffic() or {}).get("source"),
                         "kernel": "imdct_ola + qmf_analysis + sbr_core_lp + qmf_synthesis (4 launches)" if c3
                                   else "imdct_ola + qmf_analysis + sbr_core_hq + ps + 2 x qmf_synthesis (6 launches)" if c4
                                   else "imdct_ola + limiter_front + limiter_gain + limiter_apply (4 launches)" if c2l
                                   else "xaac_imdct_ola_kernel", "kernel_ms": round(kern_ms, 5),
                         "alg_bytes_per_launch": alg_bytes},
            "bit_exact_vs_oracle": checked,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_sbr(args.workload) if (c3 or c4) else cpu_baseline(limiter=c2l)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
