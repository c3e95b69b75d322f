#!/usr/bin/env python3
"""Kernel times of the round-2 additions outside the C4 chain -- the USAC FD IMDCT and the eSBR (Path A) QMF banks -- on
resident synthetic batches, HIP-event timed on the library's stream; prints one JSON line per kernel with the achieved
algorithmic GB/s against the 8 TB/s HBM roof.  Run on the GPU box: python tools/bench_new_kernels.py [n_ch]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timed(torch, ctx, fn, steps=20, warmup=3):
    for _ in range(warmup):
        fn()
    ctx.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3  # us


def main():
    import torch
    import libxaac_amd
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(0)
    res = []
    # USAC FD IMDCT: long frames and short frames
    coef = torch.from_numpy((rng.integers(-2 ** 31, 2 ** 31, (n, 1024)) >> 8).astype(np.int32)).to(dev)
    ov = torch.zeros((n, 1024), dtype=torch.int32, device=dev)
    sp = torch.zeros(n, dtype=torch.uint8, device=dev)
    out = torch.zeros((n, 1024), dtype=torch.int32, device=dev)
    for name, seq in (("usac_imdct_long", 0), ("usac_imdct_short", 2)):
        ics = torch.tensor([[seq, 1]] * n, dtype=torch.uint8, device=dev)
        us = timed(torch, ctx, lambda: ctx.usac_imdct_process_batch(coef, ics, ov, sp, out))
        res.append((name, us, n * 16384))
    # 960-line AAC IMDCT: 3840 B lines + 1920 B overlap in, 3840 B samples + 1920 B overlap out per channel-frame
    spec9 = torch.from_numpy(rng.integers(-2 ** 17, 2 ** 17, (n, 960)).astype(np.int32)).to(dev)
    ov9 = torch.zeros((n, 480), dtype=torch.int32, device=dev)
    out9 = torch.zeros(n * 960, dtype=torch.int32, device=dev)
    for name, seq in (("imdct960_long", 0), ("imdct960_short", 2)):
        ics9 = torch.tensor([[seq, 1]] * n, dtype=torch.uint8, device=dev)
        st9 = torch.tensor([[seq, 1]] * n, dtype=torch.uint8, device=dev)
        us = timed(torch, ctx, lambda: ctx.imdct960_process_batch(spec9, ics9, ov9, st9, out9))
        res.append((name, us, n * 11520))
    # AAC-LD / ELD IMDCT: lines + overlap in, PCM16 + overlap out (LD: F/2 words of overlap, ELD: 3 F in, 2.75 F out)
    for fl in (512, 480):
        for eld in (0, 1):
            nov = 3 * fl if eld else fl // 2
            specl = torch.from_numpy(rng.integers(-2 ** 17, 2 ** 17, (n, fl)).astype(np.int32)).to(dev)
            ovl = torch.zeros((n, nov), dtype=torch.int32, device=dev)
            shp = torch.zeros(n, dtype=torch.uint8, device=dev)
            spv = torch.zeros(n, dtype=torch.uint8, device=dev)
            pcml = torch.zeros(n * fl, dtype=torch.int16, device=dev)
            us = timed(torch, ctx, lambda: ctx.imdct_ld_process_batch(specl, shp, ovl, spv, pcml, fl, eld))
            res.append(("imdct_%s_%d" % ("eld" if eld else "ld", fl), us, n * (4 * fl + 2 * fl + (4 * nov + 11 * fl if eld else 8 * nov))))
    # eSBR banks
    core = torch.from_numpy((rng.uniform(-1, 1, (n, 1024)) * 20000).astype(np.float32)).to(dev)
    sa = torch.zeros((n, libxaac_amd.ESBR_ANA_STATE_WORDS), dtype=torch.int32, device=dev)
    ss = torch.zeros((n, libxaac_amd.ESBR_SYN_STATE_WORDS), dtype=torch.int32, device=dev)
    re = torch.zeros((n, 32, 64), dtype=torch.float32, device=dev)
    im = torch.zeros((n, 32, 64), dtype=torch.float32, device=dev)
    pcm = torch.zeros((n, 2048), dtype=torch.float32, device=dev)
    us = timed(torch, ctx, lambda: ctx.esbr_qmf_analysis_batch(core, sa, re, im))
    res.append(("esbr_qmf_analysis", us, n * (4096 + 2 * 1288 + 2 * 4096)))   # core in, ring in/out, 32 bands re+im out
    us = timed(torch, ctx, lambda: ctx.esbr_qmf_synthesis_batch(re, im, ss, pcm))
    res.append(("esbr_qmf_synthesis", us, n * (2 * 8192 + 2 * 5128 + 8192)))  # rows in, ring in/out, samples out
    # the whole Path A chain (analysis -> HF generator + envelope adjuster -> synthesis) on side info walked from the
    # captured HE-AAC streams, 64 distinct channel set-ups tiled over the batch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import sbr_capture as c
    from esbr_structs import new_state
    from test_esbr_core_oracle_vs_reference import make_side
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
    ws = torch.zeros(ctx.esbr_workspace_bytes(n), dtype=torch.uint8, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    import esbr_structs
    off = esbr_structs.EsbrSide.reset_flag.offset

    def steady(side):   # the first frame of a stream carries the reset flag (limiter tables are built); the rest do not
        side.view(n, -1)[:, off:off + 2] = 0
    ctx.esbr_sbr_process_batch(core, hd, fr, sd, st, pcm, ws, status)
    steady(sd)
    us = timed(torch, ctx, lambda: ctx.esbr_sbr_process_batch(core, hd, fr, sd, st, pcm, ws, status))
    assert not status.cpu().numpy().any()
    # algorithmic bytes per channel-frame: 4 KB core in, 8 KB out, state in + out (its history rows dominate)
    res.append(("esbr_sbr_chain(3 kernels)", us, n * (4096 + 8192 + 2 * st.shape[1])))
    # the same chain for HE-AACv2 streams: + float parametric stereo and the second synthesis bank
    from esbr_structs import new_ps_state
    from test_esbr_ps_oracle_vs_reference import fuzz_ps_frame
    precs = [r for r in c.read_records(os.path.join(ROOT, "tests", "golden", "sbr_hq_ps_records.bin.gz")) if r["ps"] and r["frame"].apply_processing][:64]
    hs, fs, sds, pfs = [], [], [], []
    for r in precs:
        h, f = c.Header.from_buffer_copy(bytes(r["header"])), c.Frame.from_buffer_copy(bytes(r["frame"]))
        sd = make_side(rng, h, f, [0] * 10, 0, 0, False)
        sd.reset_flag = 1
        pf = fuzz_ps_frame(rng, c.PsFrame.from_buffer_copy(bytes(r["ps_frame"])), 0)
        for x, lst in ((h, hs), (f, fs), (sd, sds), (pf, pfs)):
            lst.append(np.frombuffer(bytes(x), np.uint8))
    hd, fr, sd, pf = tile(hs), tile(fs), tile(sds), tile(pfs)
    st = torch.from_numpy(np.stack([np.frombuffer(bytes(new_state()), np.uint8)] * n)).to(dev)
    pst = torch.from_numpy(np.stack([np.frombuffer(bytes(new_ps_state()), np.uint8)] * n)).to(dev)
    pcm_r = torch.zeros((n, 2048), dtype=torch.float32, device=dev)
    ctx.esbr_sbr_process_batch(core, hd, fr, sd, st, pcm, ws, status, pf, pst, pcm_r)
    steady(sd)
    us = timed(torch, ctx, lambda: ctx.esbr_sbr_process_batch(core, hd, fr, sd, st, pcm, ws, status, pf, pst, pcm_r))
    assert not status.cpu().numpy().any()
    res.append(("esbr_sbr_ps_chain(5 kernels)", us, n * (4096 + 2 * 8192 + 2 * st.shape[1] + 2 * pst.shape[1] + pf.shape[1])))
    # the QMF-domain harmonic transposer (ixheaacd_qmf_hbe_apply without a pitch): the stream headers' bank (synth_size
    # 12, stretch factors 2 and 3) and the largest bank with all three factors; algorithmic bytes: 2 x 8 KB rows in,
    # 2 x 8 KB rows out, the state's delay lines and row buffers in + out
    from make_golden_hbe import state_from_params
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    nh = min(n, 8192)
    for label, par in (("hbe_apply(size 12, x2 x3)", [12, 6, 15, 41, 15, 29, 41, 0, 0, 0, 3]),
                       ("hbe_apply(size 8, x2 x3 x4)", [8, 2, 9, 31, 9, 15, 27, 31, 0, 0, 4])):
        hst = torch.from_numpy(np.stack([np.frombuffer(bytes(state_from_params(par)), np.uint8)] * nh)).to(dev)
        qre = torch.from_numpy((rng.standard_normal((nh, 32, 64)) * 1000).astype(np.float32)).to(dev)
        qim = torch.from_numpy((rng.standard_normal((nh, 32, 64)) * 1000).astype(np.float32)).to(dev)
        pvr, pvi = torch.zeros_like(qre), torch.zeros_like(qre)
        hstat = torch.zeros(nh, dtype=torch.int32, device=dev)
        us = timed(torch, ctx, lambda: ctx.hbe_apply_batch(qre, qim, hst, pvr, pvi, hstat))
        assert not hstat.cpu().numpy().any()
        res.append((label, us * n / nh, n * (4 * 8192 + 2 * libxaac_amd.HBE_STATE_BYTES)))
    cpu = reference_cpu_rates()
    for name, us, bytes_ in res:
        line = {"kernel": name, "n_ch": n, "us": round(us, 1), "channel_frames_per_s": round(n / us * 1e6),
                "alg_GBps": round(bytes_ / us / 1e3, 1), "frac_of_8TBps": round(bytes_ / us / 1e3 / 8000, 4)}
        if name in cpu:
            line["cpu_reference_one_core_channel_frames_per_s"] = round(cpu[name])
        print(json.dumps(line))


def reference_cpu_rates(m=4000):
    """the compiled reference's own functions (oracle/_ref/libref_harness.so, C loops in oracle/ref_harness.c) on one host
    core over m channel-frames of the same kind of input: the CPU baseline of the 960-line and LD / ELD IMDCT rows"""
    import ctypes
    import time
    so = os.path.join(ROOT, "oracle", "_ref", "libref_harness.so")
    if not os.path.exists(so):
        return {}
    lib = ctypes.CDLL(so)
    rng = np.random.default_rng(1)
    P32, P16 = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int16)
    out = {}
    base = np.zeros((m, 2048), np.int32)
    ov = np.zeros((m, 2048), np.int32)
    o32, o16 = np.zeros(1024, np.int32), np.zeros(1024, np.int16)
    for name, fl, call in (("imdct960_long", 960, lambda s: lib.ref_imdct960_loop(m, s.ctypes.data_as(P32), ov.ctypes.data_as(P32), o32.ctypes.data_as(P32))),
                           ("imdct_ld_512", 512, lambda s: lib.ref_imdct_ld_loop(m, s.ctypes.data_as(P32), ov.ctypes.data_as(P32), 512, 23, o16.ctypes.data_as(P16))),
                           ("imdct_eld_512", 512, lambda s: lib.ref_imdct_ld_loop(m, s.ctypes.data_as(P32), ov.ctypes.data_as(P32), 512, 39, o16.ctypes.data_as(P16))),
                           ("imdct_ld_480", 480, lambda s: lib.ref_imdct_ld_loop(m, s.ctypes.data_as(P32), ov.ctypes.data_as(P32), 480, 23, o16.ctypes.data_as(P16))),
                           ("imdct_eld_480", 480, lambda s: lib.ref_imdct_ld_loop(m, s.ctypes.data_as(P32), ov.ctypes.data_as(P32), 480, 39, o16.ctypes.data_as(P16)))):
        if not hasattr(lib, "ref_imdct960_loop"):
            return {}
        best = 1e9
        for _ in range(3):
            spec = base.copy()
            spec[:, :fl] = rng.integers(-2 ** 17, 2 ** 17, (m, fl))
            ov[:] = 0
            t0 = time.perf_counter()
            call(spec)
            best = min(best, time.perf_counter() - t0)
        out[name] = m / best
    return out


if __name__ == "__main__":
    main()
