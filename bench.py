#!/usr/bin/env python3
"""bench.py -- decoded audio frames/s of the MI355X transform back-end.

Default workload = the configuration BASELINE.json's metric is quoted on (configs[3], SURVEY.md §8d "C4"):
HE-AACv2 48 kHz, batch = 8192 streams per step, one frame each: mono 1024-sample IMDCT + window/overlap-add
(PCM16 hand-off of the SBR case) -> complex 32-band QMF analysis -> LPP transposer + envelope adjustment
(HQ SBR) -> parametric stereo (hybrid filterbank, decorrelation, rotation) -> two complex 64-band QMF
synthesis banks -> 2048 L,R PCM16 pairs.  Synthetic core spectra (seeded, uniform in +-2^17 below bin 512),
SBR / PS side info cycled from reference-captured HE-AACv2 frames (tests/golden/sbr_hq_ps_records.bin.gz), all
per-stream state (IMDCT overlap, SBR, PS, filterbank rings) carried in HBM.  Each rank owns `--sets` x 8192
independent streams and decodes one frame of 8192 of them per step, round-robin, so the working set streams from
HBM rather than from the Infinity Cache.  With --gpus N every rank runs this on its own 8192 x sets streams
(C5: weak scaling, streams sharded by rank, no data-path collective); after the timed region the ranks' PCM is
gathered once over RCCL (the north star's "final interleaved gather") and that time is reported separately.

A "step" = one pass of the hot path over one batch.  Inputs are resident in HBM before the timed region.
`value` = frames decoded by all ranks / wall time (max over ranks).  `roofline.achieved` = algorithmic bytes per
step / mean step duration from HIP events recorded on the launch stream.  Every run checks itself
(outside the timed region): the status words of the last step must all be zero, and one more step of a
256-stream slice is decoded by the oracle chain on the host from the same device state and compared word for
word (`bit_exact_vs_oracle`).  `secondary` (N=1 only; `f4_transforms` = the USAC / 960-line / LD / ELD IMDCT kernels) = short runs of C2 (AAC-LC IMDCT, BASELINE configs[1]), C4A (the
same HE-AACv2 streams through the reference's default float eSBR path) and
C3 (HE-AACv1, configs[2]) in the same process.  `cpu_baseline` = the same workload on the host cores (the
compiled reference when oracle/_ref travelled with the repo, else the bit-exact restatement), bounded sample,
rank 0 at N=1 only.

  python bench.py --gpus 1 --steps 100 --warmup 10
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
# multi-process GPU work on this driver needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); read at runtime start-up,
# so it is set before torch is imported, whichever way the ranks were started
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

FRAMES_PER_STEP = 8192          # stereo frames per batch (BASELINE configs[1])
CH = 2
ALG_BYTES_PER_FRAME = 20480     # R spec 2x4096 + ovl 2x2048, W pcm16 2x2048 + ovl 2x2048
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_latest.json")       # tools/pmc_to_json.py, from the rocprofv3 --pmc passes
PMC_FILE_C2 = os.path.join(ROOT, "profiles", "pmc_hbm_latest.json")   # round-1 C2 counters (that kernel has not changed)
VALU_CYCLES_PER_WAVE_INSTR = 4.4   # measured: tools/ubench/valu_rate.hip, profiles/r01_m_c2_sq_detail.txt (v_mul_hi_i32 = a shift)
SIMDS, CLOCK_GHZ = 1024, 2.4       # 256 CUs x 4 SIMDs; MI355X_MICROARCH.md peak engine clock
PREWARM_STEPS = 60                 # untimed steps before the counted warm-up (Workload.run; profiles/r06_a_clock_probe.txt)


def library_build_id():
    """the id xaac_version() carries: sha256 of the sources the loaded library was built from (csrc/Makefile: BUILD_ID)"""
    try:
        import libxaac_amd
        v = libxaac_amd.load_library().xaac_version().decode()
        return v.split(" build ")[1] if " build " in v else None
    except Exception:
        return None


def measured_traffic(workload):
    """HBM bytes per step from the last committed PMC profile of this workload's kernels (rocprofv3 FETCH_SIZE
    x2 + WRITE_SIZE summed over the chain's launches, calibrated as MI355X_MICROARCH.md prescribes), or {}.  The C4
    profile names the build id of the library it was taken on (tools/pmc_to_json.py); a line printed by another build
    reports traffic null and says why."""
    try:
        if workload == "c4":
            with open(PMC_FILE) as f:
                d = json.load(f)
            have, want = library_build_id(), d.get("build_id")
            if want is None or have != want:
                return {"bytes_per_step": None,
                        "source": "%s holds counters of library build %s; this run's library is build %s: not this library's "
                                  "traffic" % (os.path.relpath(PMC_FILE, ROOT), want, have)}
            return {"bytes_per_step": d["c4"]["bytes_per_step"], "source": d["source"] + " (library build %s)" % want}
        with open(PMC_FILE_C2) as f:
            return json.load(f).get(workload) or {}
    except (OSError, ValueError, KeyError):
        return {}


def valu_roofline(workload, kernel_ms):
    """The chain's other bound: wave-instructions issued per step from the committed SQ counters, the time
    the chip's 1024 SIMDs need to issue them at the measured 4.4 cycles per wave64 instruction, and how much of the
    measured kernel time that is.  None without counters for the workload."""
    if workload != "c4":
        return None
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
        c = d["c4"]
    except (OSError, ValueError, KeyError):
        return None
    to_ms = lambda n: n / SIMDS * VALU_CYCLES_PER_WAVE_INSTR / (CLOCK_GHZ * 1e9) * 1e3
    floor_valu, floor_all = to_ms(c["valu_wave_instr"]), to_ms(c["valu_wave_instr"] + c["salu_wave_instr"] + c["lds_wave_instr"])
    lane_ops = int(sum(v.get("valu", 0) * v.get("active_lanes", 0) for v in d["kernels"].values() if v.get("active_lanes")))
    return {"valu_wave_instr": c["valu_wave_instr"], "salu_wave_instr": c["salu_wave_instr"], "lds_wave_instr": c["lds_wave_instr"],
            "cycles_per_wave_instr": VALU_CYCLES_PER_WAVE_INSTR, "simds": SIMDS, "clock_ghz": CLOCK_GHZ,
            "issue_floor_ms_valu": round(floor_valu, 4), "issue_floor_ms_all": round(floor_all, 4),
            "issue_frac": round(floor_valu / kernel_ms, 3) if kernel_ms else None,
            "issue_frac_if_salu_and_lds_issued_serially": round(floor_all / kernel_ms, 3) if kernel_ms else None,
            "lane_ops_per_step": lane_ops,
            "issue_floor_ms_valu_at_full_lanes": round(to_ms(lane_ops / 64.0), 4) if lane_ops else None,
            "active_lanes_per_valu_instr": {k: v.get("active_lanes") for k, v in d["kernels"].items() if "active_lanes" in v},
            "note": "issue_frac = VALU-only floor / measured kernel time: the time the chip's SIMDs need to issue this step's "
                    "VALU wave-instructions at 4.4 cycles each (32-bit multiplies issue like adds and shifts).  SALU and LDS "
                    "instructions issue on their own ports beside another wave's VALU; charging them 4.4 cycles each as well "
                    "gives the second, pessimistic figure.  ..._at_full_lanes = the same arithmetic with all 64 lanes busy",
            "source": d["source"]}


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
    # What a host that parsed these streams knows about them (xaac_sbr_hq_batch.max_band_hint): no band number of any header, frame
    # or starting state reaches above QMF band 48 -- 24 kHz cores with the SBR range ending at or below band 48 -- so the pass that
    # takes other streams through 64-band rows has nothing to do and its (empty) launch is left out.  A stream for which this were
    # wrong would be refused and the run's own status check (every status word zero) would fail.
    def top_band(r):
        h, f, st = (cap.Header.from_buffer_copy(bytes(r["header"])), cap.Frame.from_buffer_copy(bytes(r["frame"])),
                    cap.State.from_buffer_copy(bytes(r["st0"])))
        t = [h.sub_band_end, f.max_qmf_subband_aac, st.syn_usb, st.syn_lsb, st.codec_usb, st.prev_max_qmf_subband_aac]
        t += list(h.freq_band_tbl_hi[:h.num_sf_bands[1] + 1]) + list(h.freq_band_tbl_lo[:h.num_sf_bands[0] + 1])
        t += [max(q.src_end_band + q.dst_end_band, q.dst_start_band + q.num_bands_in_patch) for q in h.patch[:h.num_patches]]
        return max(t)
    hint = 48 if max(top_band(r) for grp in groups for r in grp) <= 48 else 0
    batches = []
    for s in range(sets):
        spec = torch.randint(-(1 << 17), 1 << 17, (n, 1024), generator=g, device=device, dtype=torch.int32)
        spec[:, 512:] = 0
        batches.append({"max_band_hint": hint, "spec": spec, "ics": torch.zeros((n, 2), dtype=torch.uint8, device=device),
                        "overlap": torch.zeros((n, 512), dtype=torch.int32, device=device),
                        "state": torch.zeros((n, 2), dtype=torch.uint8, device=device),
                        "core_pcm": torch.zeros(n * 1024, dtype=torch.int16, device=device),
                        "hdr": hdr, "frames": frames, "sbr_state": st0.clone(), "ps_state": ps0.clone(),
                        "pcm": torch.zeros(n * 4096, dtype=torch.int16, device=device)})
    return batches


_LANES = {}


def extra_lanes(torch, libxaac_amd, dev, n):
    """n further (HIP stream, context) pairs on dev, the same ones every time: every leg of a run that deals its steps out over
    several streams uses these.  (A stream per leg left the process with half a dozen streams by the time the end-to-end leg ran,
    and that leg's own copy and compute streams then shared hardware queues: 3.1 -> 2.6 x 10^6 frames/s on the same box.)"""
    key = (dev.index or 0)
    have = _LANES.setdefault(key, [])
    while len(have) < n:
        st = torch.cuda.Stream(device=dev)
        have.append((st, libxaac_amd.XaacContext(dev.index or 0, st.cuda_stream)))
    return have[:n]


def secondary_f4(torch, libxaac_amd, ctx, dev, n=16384, launches=20, hip_streams=2):
    """SURVEY.md row f4's transforms outside the C4 chain -- USAC FD, 960-line, AAC-LD and AAC-ELD IMDCT -- on resident synthetic
    batches of n channel-frames: microseconds per launch (wall clock around `launches` back-to-back launches on the
    library's stream) and the fraction of the HBM roof their algorithmic bytes reach.  Parity: tests/test_usac_imdct.py,
    tests/test_imdct960_gpu.py, tests/test_imdct_ld_gpu.py."""
    import numpy as np
    rng = np.random.default_rng(4)
    out = {}

    # launches dealt out over `hip_streams` HIP streams like the headline's steps: lane q = its own context and its own carried
    # buffers (overlap, window state, outputs: `fresh` clones them), an independent batch of n channel-frames; inputs are shared
    ctxs = [ctx] + [c for _, c in extra_lanes(torch, libxaac_amd, dev, hip_streams - 1)]
    fresh = lambda *ts: [ts] + [tuple(t.clone() for t in ts) for _ in ctxs[1:]]

    def timed(name, fn, alg_bytes, nl=None):
        """fn(q): one launch on lane q; nl: lanes used (default: all)"""
        nl = len(ctxs) if nl is None else nl
        for i in range(3 * nl):
            fn(i % nl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(launches):
            fn(i % nl)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / launches * 1e6
        out[name] = {"us_per_launch": round(us, 1), "channel_frames_per_s": round(n / us * 1e6),
                     "roofline_frac": round(alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "hip_streams": nl}

    spec = torch.from_numpy(rng.integers(-2 ** 17, 2 ** 17, (n, 1024)).astype(np.int32)).to(dev)
    z8 = lambda *shape: torch.zeros(shape, dtype=torch.uint8, device=dev)
    ov = torch.zeros((n, 1024), dtype=torch.int32, device=dev)
    o32 = torch.zeros((n, 1024), dtype=torch.int32, device=dev)
    ics_u = torch.tensor([[0, 1]] * n, dtype=torch.uint8, device=dev)
    sp_u = z8(n)
    lu = fresh(ov, sp_u, o32)
    timed("usac_fd_1024", lambda q: ctxs[q].usac_imdct_process_batch(spec, ics_u, lu[q][0], lu[q][1], lu[q][2]), n * 16384)
    spec9 = spec[:, :960].contiguous()
    ov9 = torch.zeros((n, 480), dtype=torch.int32, device=dev)
    out9 = torch.zeros(n * 960, dtype=torch.int32, device=dev)
    ics9, st9 = z8(n, 2), z8(n, 2)
    l9 = fresh(ov9, st9, out9)
    timed("aac_960", lambda q: ctxs[q].imdct960_process_batch(spec9, ics9, l9[q][0], l9[q][1], l9[q][2]), n * 11520)
    for fl in (512, 480):
        for eld in (0, 1):
            nov = 3 * fl if eld else fl // 2
            sl = spec[:, :fl].contiguous()
            ol = torch.zeros((n, nov), dtype=torch.int32, device=dev)
            shp, spv = z8(n), z8(n)
            pcm = torch.zeros(n * fl, dtype=torch.int16, device=dev)
            ll = fresh(ol, spv, pcm)
            timed("%s_%d" % ("aac_eld" if eld else "aac_ld", fl),
                  lambda q, sl=sl, ll=ll, shp=shp, fl=fl, eld=eld: ctxs[q].imdct_ld_process_batch(sl, shp, ll[q][0], ll[q][1], ll[q][2], fl, eld),
                  n * (4 * fl + 2 * fl + (4 * nov + 11 * fl if eld else 8 * nov)))
    out["n_channel_frames"] = n
    out["timing"] = ("wall clock around back-to-back launches dealt out over %d HIP streams (launch overhead included): roofline_frac is on "
                     "the step's own clock" % len(ctxs))
    # the PVC envelope decoder (tests/test_pvc.py): 2:1 frames at start band 12, the QMF rows of one stream-frame per channel
    fr = np.zeros((n, libxaac_amd.PVC_FRAME_BYTES), np.uint8)
    fr[:, 0], fr[:, 2], fr[:, 4] = 1 + (np.arange(n) & 1), 2, 12           # pvc_mode, pvc_rate, first_bnd_idx
    fr[:, 8:40:2] = rng.integers(0, 128, (n, 16)).astype(np.uint8)        # pvc_id (little-endian uint16)
    pv_f = torch.from_numpy(fr).to(dev)
    pv_q = torch.randn((2048, 64, 64), dtype=torch.float32, device=dev).mul_(900.0).repeat(n // 2048, 1, 1)
    pv_st, pv_out = z8(n, libxaac_amd.PVC_STATE_BYTES), torch.zeros((n, 16, 64), dtype=torch.float32, device=dev)
    lp = fresh(pv_st, pv_out)
    timed("esbr_pvc", lambda q: ctxs[q].pvc_process_batch(pv_f, pv_q, pv_q, lp[q][0], lp[q][1]), n * (32 * 16 * 2 * 4 + 228 + 4096 + 188))
    return out


def secondary_end_to_end(copies=4096):
    """The repo's own decoder on real streams, end to end: `copies` HE-AACv2 ADTS streams (the committed
    tests/golden/streams/mix_aot29_32k.aac, 38 frames each) decoded in lock step by libxaac_amd/decoder.py -- host parser
    threads (libxaac_host.so, no reference code) -> pinned staging -> H2D (spectra + window info + SBR / PS side info only)
    -> IMDCT + HQ SBR + PS on device-resident state -> D2H PCM -- with the parse of step k + 1 overlapping the GPU's step k.
    The PCIe-inclusive, parser-inclusive rate; the host parser alone on the same machine beside it (the ceiling of a host
    that parses on these CPU cores), and the byte-exact check against the committed CRC of the reference decoder's PCM."""
    import zlib
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_decoder
    from libxaac_amd import decoder
    name = "mix_aot29_32k"
    data = open(os.path.join(ROOT, "tests", "golden", "streams", name + ".aac"), "rb").read()
    pcm, rate = decoder.decode_streams([data] * 4)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "decoder_ref.npz"))
    k = 4   # index of mix_aot29_32k in tools/make_golden_parser.py NAMES
    exact = all(zlib.crc32(np.ascontiguousarray(p).tobytes()) & 0xffffffff == int(gold["crc"][k]) for p in pcm)
    best = None
    for _ in range(3):
        timing = {}
        decoder.decode_streams([data] * copies, keep_pcm=False, timing=timing)
        if best is None or timing["steps_s"] < best["steps_s"]:
            best = timing
    parser = 0.0
    for _ in range(3):   # (a pause in front of each try: the decode runs above have used up the cgroup's CPU quota of their periods)
        time.sleep(0.25)
        parser = max(parser, bench_decoder.parser_only(decoder, data, copies, 0))
    # the same streams decoded the way the reference does with its default flags (-esbr:1: float eSBR tools, the QMF harmonic
    # transposer on every frame, float PS; decode_streams(esbr=True)), checked against the CRC of `xaacdec`'s PCM
    pcm_e, _ = decoder.decode_streams([data] * 4, esbr=True)
    exact_e = all(zlib.crc32(np.ascontiguousarray(p).tobytes()) & 0xffffffff == int(gold["crc_esbr"][k]) for p in pcm_e)
    best_e = None
    for _ in range(2):
        timing = {}
        decoder.decode_streams([data] * copies, keep_pcm=False, timing=timing, esbr=True)
        if best_e is None or timing["steps_s"] < best_e["steps_s"]:
            best_e = timing
    native = native_esbr = None
    cli = os.path.join(ROOT, "libxaac_amd", "xaacdec_amd")
    if os.path.exists(cli):   # the same loop without Python: libxaac_amd/host/xaacdec_amd.cpp (HIP runtime + the two libraries)
        import subprocess
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            for flag in ("-esbr:0", "-esbr:1"):   # the reference's two readings of the SBR payload (its default is -esbr:1)
                r = subprocess.run([cli, "-ifile:" + os.path.join(ROOT, "tests", "golden", "streams", name + ".aac"),
                                    "-ofile:" + os.path.join(tmp, "o.wav"), "-copies:%d" % copies, flag], capture_output=True, text=True,
                                   timeout=300)
                if r.returncode == 0:
                    j = json.loads(r.stdout.strip().splitlines()[-1])
                    res = {"frames_per_s_after_first_step": j["frames_per_s_after_first_step"], "frames_per_s_whole_run": j["frames_per_s"],
                           "wall_s": j["wall_s"]}
                    if flag == "-esbr:0":
                        native = res
                    else:
                        native_esbr = res
    # the reference decoder itself on this machine's cores, end to end on the same stream (repeated into a longer file so
    # that process start-up does not count): P processes of oracle/_ref/xaacdec -esbr:0 side by side
    ref_cpu = None
    xaacdec = os.path.join(ROOT, "oracle", "_ref", "xaacdec")
    if os.path.exists(xaacdec):
        import subprocess
        import tempfile
        procs, reps = min(os.cpu_count() or 1, 128), 60
        with tempfile.TemporaryDirectory() as tmp:
            long_aac = os.path.join(tmp, "long.aac")
            open(long_aac, "wb").write(data * reps)
            t0 = time.perf_counter()
            ps = [subprocess.Popen([xaacdec, "-ifile:" + long_aac, "-ofile:" + os.path.join(tmp, "o%d.wav" % i), "-esbr:0"],
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i in range(procs)]
            ok = all(p.wait() == 0 for p in ps)
            dt = time.perf_counter() - t0
        if ok:
            ref_cpu = {"value": round(procs * reps * 38 / dt, 1), "unit": "frames/s", "processes": procs, "kind": "reference",
                       "sample": "%d processes of oracle/_ref/xaacdec -esbr:0, each on the stream repeated %d times (%d frames)" % (procs, reps, reps * 38)}
    return {"metric": "HE-AACv2 ADTS streams decoded end to end (own host parser + GPU, PCIe inclusive)",
            "value": round(best["frames"] / best["steps_s"], 1), "unit": "frames/s", "streams": copies, "frames": best["frames"],
            "parse_s": round(best["parse_s"], 4), "gpu_and_copies_s": round(best["gpu_s"], 4), "wall_s": round(best["steps_s"], 4),
            "parser_only_frames_per_s": round(parser, 1), "host_threads": os.cpu_count(),
            "pcm_equals_reference_decoder": exact, "stream": name + ".aac", "output_rate_hz": rate,
            "native_cli": native, "reference_decoder_on_host_cores": ref_cpu,
            "esbr": {"what": "the same streams with the reference's default flags (-esbr:1, Path A) through decode_streams(esbr=True)",
                     "value": round(best_e["frames"] / best_e["steps_s"], 1), "unit": "frames/s", "wall_s": round(best_e["steps_s"], 4),
                     "parse_s": round(best_e["parse_s"], 4), "gpu_and_copies_s": round(best_e["gpu_s"], 4),
                     "pcm_equals_reference_decoder": exact_e, "native_cli": native_esbr}}


def end_to_end_in_its_own_process():
    """secondary_end_to_end() in a fresh interpreter: the decoder's loop has its own three HIP streams (copies up, kernels, copies
    down), and in a process that has already driven two compute streams they lose a fifth of their rate to each other (measured on
    one box, same library: 3.1 x 10^6 frames/s behind a one-stream run of the legs above, 2.6-2.8 behind a two-stream run) -- a
    property of this process's history, not of the decoder, which a host would run in a process of its own anyway."""
    import subprocess
    code = "import json, bench; print('E2E_JSON ' + json.dumps(bench.secondary_end_to_end()))"
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    for line in p.stdout.splitlines()[::-1]:
        if line.startswith("E2E_JSON "):
            out = json.loads(line[len("E2E_JSON "):])
            out["process"] = "its own (a fresh interpreter started by bench.py)"
            return out
    raise RuntimeError("end-to-end subprocess: rc %d, %s" % (p.returncode, p.stderr[-300:]))


def secondary_round6(torch, libxaac_amd, ctx, dev, steps=20, n=8192):
    """The rows built in round 6, each on the step's own clock (one HIP stream, wall time over back-to-back steps): USAC channels
    at 8:3 and 4:1 SBR through xaac_esbr_sbr_process_batch (sbr_ratio), AAC-ELD's low-delay SBR call through
    xaac_sbr_eld_process_batch, the USAC FD frame with its forward-aliasing-cancellation signal made on the device (fac_in).
    Side info, states and inputs are tiled from the committed reference-made chain fixtures (first step of every chain, the
    fixtures' own input generators), so every set-up is one the reference decoder produced; the first step of every distinct
    set-up is compared with the fixture's reference CRCs."""
    import zlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff
    out = {"n_channel_frames": n, "timing": "wall clock over %d back-to-back steps on one HIP stream (launch overhead included)" % steps}

    def timed(step):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    def entry(ms, alg, ok, refused, what):
        return {"ms_per_step": round(ms, 4), "channel_frames_per_s": round(n / ms * 1e3, 1), "alg_bytes_per_step": int(alg),
                "roofline_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "first_step_equals_reference_crc": ok,
                "refused_frac": refused, "workload": what}

    # -- USAC channels at 8:3 / 4:1 SBR (tests/golden/esbr_ratio_chains.npz) ------------------------------------------------
    try:
        from make_golden_esbr_chains import chain_core
        CH = np.load(os.path.join(ROOT, "tests", "golden", "esbr_ratio_chains.npz"))
        first = [np.nonzero(CH["step_chain"] == c)[0][0] for c in range(len(CH["chain_len"]))]
        for ratio, name in ((1, "usac_esbr_8_3"), (2, "usac_esbr_4_1")):
            cs = [c for c in range(len(first)) if int(CH["chain_ratio"][c]) == ratio]
            rows = [first[c] for c in cs]
            tile = lambda a: torch.from_numpy(np.ascontiguousarray(np.stack([a[i % len(a)] for i in range(n)]))).to(dev)
            hd, fr, sd, pvs = (tile(CH[k][rows]) for k in ("header", "frame", "side", "pvc_side"))
            st, pv = tile(CH["est0"][cs]), tile(CH["pvst0"][cs])
            core = tile(np.stack([chain_core(int(CH["chain_run"][c]), int(CH["chain_id"][c]), 0) for c in cs]))
            width = 4096 if ratio == 2 else 2048
            o = torch.zeros((n, width), dtype=torch.float32, device=dev)
            status = torch.zeros(n, dtype=torch.int32, device=dev)
            ws = torch.zeros(ctx.esbr_workspace_bytes(n, ratio), dtype=torch.uint8, device=dev)
            step = lambda: ctx.esbr_sbr_process_batch(core, hd, fr, sd, st, o, ws, status, pvc_side=pvs, pvc_state=pv, sbr_ratio=ratio)
            step()
            torch.cuda.synchronize()
            og = o[:len(cs)].cpu().numpy()
            ok = bool(all(crc(og[j]) == int(CH["crc"][rows[j]][0]) for j in range(len(cs))))
            ms = timed(step)
            refused = float(status.cpu().numpy().astype(bool).mean())
            alg = n * (4 * (768 if ratio == 1 else 1024) + 4 * width + 2 * st.shape[1] + 2 * pv.shape[1] + hd.shape[1] + fr.shape[1]
                       + sd.shape[1] + pvs.shape[1])
            out[name] = entry(ms, alg, ok, refused, "%d USAC channel-frames a step, %s: %d-channel analysis bank -> float HF generator / "
                              "envelope adjuster (PVC frames: the PVC decoder inside) -> 64-band synthesis bank%s; %d distinct "
                              "reference-made set-ups tiled" % (n, "8:3 SBR, 768-sample core frames" if ratio == 1 else "4:1 SBR, 64 QMF slots",
                                                                24 if ratio == 1 else 16, " in two runs" if ratio == 2 else "", len(cs)))
    except Exception as e:
        out["usac_esbr_ratios"] = "unavailable: %r" % (e,)
    # -- AAC-ELD low-delay SBR (tests/golden/sbr_eld_chains.npz) ----------------------------------------------------------------
    try:
        from make_golden_sbr_chains import chain_pcm
        CE = np.load(os.path.join(ROOT, "tests", "golden", "sbr_eld_chains.npz"))
        cs = [c for c in range(len(CE["n_slots"])) if int(CE["n_slots"][c]) == 16]
        rows = [np.nonzero(CE["step_chain"] == c)[0][0] for c in cs]
        tile = lambda a: torch.from_numpy(np.ascontiguousarray(np.stack([a[i % len(a)] for i in range(n)]))).to(dev)
        hd, fr, st = tile(CE["header"][rows]), tile(CE["frame"][rows]), tile(CE["st0"][cs])
        pin = tile(np.stack([chain_pcm(3, c, 0)[:512] for c in cs])).reshape(-1)
        o = torch.zeros(n * 1024, dtype=torch.int16, device=dev)
        status = torch.zeros(n, dtype=torch.int32, device=dev)
        ws = torch.zeros(ctx.sbr_eld_workspace_bytes(n), dtype=torch.uint8, device=dev)
        step = lambda: ctx.sbr_eld_process_batch(pin, hd, fr, st, o, ws, 16, status=status)
        step()
        torch.cuda.synchronize()
        og = o.view(n, 1024)[:len(cs)].cpu().numpy()
        ok = bool(all(crc(og[j]) == int(CE["crc"][rows[j]][0]) for j in range(len(cs))))
        ms = timed(step)
        alg = n * (2 * 512 + 2 * 1024 + 2 * st.shape[1] + hd.shape[1] + fr.shape[1])
        out["aac_eld_sbr_512"] = entry(ms, alg, ok, float(status.cpu().numpy().astype(bool).mean()),
                                       "%d AAC-ELD channel-frames a step (512-sample core frames, 16 QMF slots): LD complex analysis bank -> "
                                       "fixed-point LD-SBR core -> LD complex synthesis bank (3 launches); %d distinct reference-made "
                                       "set-ups tiled" % (n, len(cs)))
    except Exception as e:
        out["aac_eld_sbr_512"] = "unavailable: %r" % (e,)
    # -- USAC FD frames behind an LPD frame, every one with FAC data: the signal made on the device ----------------------------------
    try:
        rng = np.random.default_rng(11)
        ccfl = 1024
        fin = np.zeros((n, libxaac_amd.USAC_FAC_IN_WORDS), np.int32)
        fin[:, 0] = rng.integers(0, 120, n)
        fin[:, 1:129] = rng.integers(-40, 41, (n, 128)) * (rng.integers(0, 4, (n, 128)) == 0)
        lpc = np.zeros((n, 17), np.float32)
        lpc[:, 0] = 1.0
        lpc[:, 1:] = (rng.standard_normal((n, 16)) * 0.4 * 0.8 ** np.arange(16)).astype(np.float32)
        fin[:, 129:146] = lpc.view(np.int32)
        fin[:, 146:] = (rng.standard_normal((n, 256)) * 100.0).astype(np.float32).view(np.int32)
        coef = torch.from_numpy(rng.integers(-(1 << 20), 1 << 20, (n, ccfl)).astype(np.int32)).to(dev)
        ics = torch.zeros((n, 2), dtype=torch.uint8, device=dev)
        ics[:, 0] = torch.from_numpy(np.array([3, 2, 4, 3], np.uint8)[np.arange(n) % 4]).to(dev)   # LONG_STOP, EIGHT_SHORT, STOP_START
        ov = torch.zeros((n, ccfl), dtype=torch.int32, device=dev)
        sp = torch.zeros(n, dtype=torch.uint8, device=dev)
        o32 = torch.zeros((n, ccfl), dtype=torch.int32, device=dev)
        status = torch.zeros(n, dtype=torch.int32, device=dev)
        work = torch.zeros((n, 257), dtype=torch.int32, device=dev)
        t_fin = torch.from_numpy(fin).to(dev)
        res = {}
        for label, flags_v, kw in (("fd_behind_lpd", 1, {}), ("fd_behind_lpd_with_fac_on_device", 3, {"fac_in": t_fin, "fac_work": work})):
            flags = torch.full((n,), flags_v, dtype=torch.uint8, device=dev)
            step = lambda: ctx.usac_imdct_process_batch(coef, ics, ov, sp, o32, None, status, ccfl=ccfl, lpd_flags=flags, **kw)
            res[label] = round(timed(step), 4)
            res[label + "_refused_frac"] = float(status.cpu().numpy().astype(bool).mean())
        res["workload"] = ("%d USAC FD channel-frames a step, every one behind an LPD frame; second figure: every one with FAC data and "
                           "ixheaacd_cal_fac_data on the device (xaac_usac_fac_kernel, one wave per frame) in front of the transform -- in a "
                           "stream a frame in hundreds is such a frame" % n)
        out["usac_fd_fac_ms_per_step"] = res
    except Exception as e:
        out["usac_fd_fac_ms_per_step"] = "unavailable: %r" % (e,)
    # -- the DFT harmonic transposer (-esbr_hq:1; tests/golden/hbe_dft_ref.npz) -----------------------------------------------------
    try:
        import ctypes
        import test_hbe_dft as td
        G = np.load(td.GOLDEN)
        cases = [int(c) for c in G["cases"]]
        base = [td.golden_case(G, c) for c in cases]
        m = n // 2                                    # 20 KB of state a channel
        tile = lambda rows: torch.from_numpy(np.ascontiguousarray(np.stack([rows[i % len(rows)] for i in range(m)]))).to(dev)
        st = tile([np.frombuffer(bytes(b[0]), np.uint8) for b in base])
        cfg_tab = torch.from_numpy(np.stack([np.frombuffer(bytes(b[1]), np.uint8) for b in base])).to(dev)
        cre = torch.from_numpy(np.stack([b[2][0] for b in base])).to(dev)
        cim = torch.from_numpy(np.stack([b[2][1] for b in base])).to(dev)
        cfg_idx = torch.tensor([i % len(cases) for i in range(m)], dtype=torch.int32, device=dev)
        rngs = [np.random.default_rng(4000 + c) for c in cases]
        ins = [td.golden_inputs(c, 0, r) for c, r in zip(cases, rngs)]
        qre, qim = tile([i[0][0] for i in ins]), tile([i[0][1] for i in ins])
        pvr, pvi = tile([i[1][0] for i in ins]), tile([i[1][1] for i in ins])
        ovs = torch.tensor([ins[i % len(cases)][2] for i in range(m)], dtype=torch.int32, device=dev)
        pitch = torch.tensor([ins[i % len(cases)][3] for i in range(m)], dtype=torch.int32, device=dev)
        status = torch.zeros(m, dtype=torch.int32, device=dev)
        step = lambda: ctx.hbe_dft_apply_batch(qre, qim, cfg_tab, cre, cim, st, pvr, pvi, status, pitch_in_bins=pitch, oversampling=ovs, cfg=cfg_idx)
        step()
        torch.cuda.synchronize()
        got = pvr[:len(cases)].cpu().numpy()
        ok = True
        for j, c in enumerate(cases):                 # float path: the reference's rows to 2e-5 of their peak (include/xaac_hbe.h)
            L, a0 = base[j][0].anal.analy_size, base[j][0].anal.a_start
            ref = G["rows_%d" % c][0, 0]
            ok = ok and bool(np.abs(got[j][:, a0:a0 + L] - ref).max() <= 2e-5 * np.abs(ref).max())
        ms = timed(step)
        alg = m * (2 * 2048 * 4 + 2 * 32 * 32 * 4 + 2 * (st.shape[1] - 4 * 1280))
        e = entry(ms, alg, ok, float(status.cpu().numpy().astype(bool).mean()),
                  "%d channel-frames a step through ixheaacd_dft_hbe_apply (xaac_hbe_dft_apply_batch_run: synthesis bank, eight hops of "
                  "real FFT / polar stretch by 2..4 / inverse FFT / overlap-add, analysis bank); %d reference-made configurations "
                  "(synth_size 8 / 12 / 16, analy_size 28 / 32, with and without oversampling and pitch) tiled" % (m, len(cases)))
        e["channel_frames_per_s"] = round(m / ms * 1e3, 1)
        e["first_step_within_2e-5_of_reference"] = e.pop("first_step_equals_reference_crc")
        out["hbe_dft_transposer"] = e
    except Exception as e:
        out["hbe_dft_transposer"] = "unavailable: %r" % (e,)
    return out


def secondary_esbr(torch, libxaac_amd, ctx, dev, steps, warmup, hip_streams=2):
    """The same HE-AACv2 streams through the reference's DEFAULT SBR path (-esbr:1, "Path A": 32-bit-ring QMF banks, float
    LPP transposer / envelope adjuster / parametric stereo; docs/NOTEBOOK.md 5f): xaac_esbr_sbr_process_batch on float core
    samples, 8192 streams per step, side info tiled from 64 reference-captured HE-AACv2 frames with synthetic float
    envelope data.  One frame step of the 64 distinct set-ups is compared word for word with the oracle chain."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sbr_capture as cap
    from esbr_structs import new_state, new_ps_state
    from test_esbr_core_oracle_vs_reference import make_side
    from test_esbr_ps_oracle_vs_reference import fuzz_ps_frame
    n = FRAMES_PER_STEP
    rng = np.random.default_rng(7)
    recs = [r for r in cap.read_records(os.path.join(ROOT, "tests", "golden", "sbr_hq_ps_records.bin.gz"))
            if r["ps"] and r["frame"].apply_processing][:64]
    hs, fs, sds, pfs = [], [], [], []
    for r in recs:
        h, f = cap.Header.from_buffer_copy(bytes(r["header"])), cap.Frame.from_buffer_copy(bytes(r["frame"]))
        sd = make_side(rng, h, f, [0] * 10, 0, 0, False)
        sd.reset_flag = 1
        hs.append(h), fs.append(f), sds.append(sd)
        pfs.append(fuzz_ps_frame(rng, cap.PsFrame.from_buffer_copy(bytes(r["ps_frame"])), 0))
    tile = lambda xs: torch.from_numpy(np.stack([np.frombuffer(bytes(xs[i % len(xs)]), np.uint8) for i in range(n)])).to(dev)
    hd, fr, sd, pf = tile(hs), tile(fs), tile(sds), tile(pfs)
    st0 = torch.from_numpy(np.stack([np.frombuffer(bytes(new_state()), np.uint8)] * n)).to(dev)
    ps0 = torch.from_numpy(np.stack([np.frombuffer(bytes(new_ps_state()), np.uint8)] * n)).to(dev)
    st, pst = st0.clone(), ps0.clone()
    core_h = (rng.uniform(-1, 1, (n, 1024)) * 12000.0).astype(np.float32)
    core = torch.from_numpy(core_h).to(dev)
    ws = torch.zeros(ctx.esbr_workspace_bytes(n), dtype=torch.uint8, device=dev)
    out_l = torch.zeros((n, 2048), dtype=torch.float32, device=dev)
    out_r = torch.zeros((n, 2048), dtype=torch.float32, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    # the timed steps are dealt out over `hip_streams` HIP streams like the headline's (Workload): lane q = its own context,
    # states, workspace and outputs (an independent batch of the same streams); the side info and the core samples are shared
    lanes = [(ctx, st, pst, ws, out_l, out_r, status)]
    for _, cq in extra_lanes(torch, libxaac_amd, dev, hip_streams - 1):
        lanes.append((cq, st0.clone(), ps0.clone(), torch.zeros_like(ws), torch.zeros_like(out_l), torch.zeros_like(out_r),
                      torch.zeros_like(status)))
    run_on = lambda q, **kw: lanes[q][0].esbr_sbr_process_batch(core, hd, fr, sd, lanes[q][1], lanes[q][4], lanes[q][3], lanes[q][6],
                                                               pf, lanes[q][2], lanes[q][5], **kw)
    run = lambda: run_on(0)
    for q in range(len(lanes)):
        run_on(q)     # every lane's first frame comes with the reset flag
    torch.cuda.synchronize()
    refused = float(status.cpu().numpy().astype(bool).mean())
    try:  # the first step (fresh states) against the oracle, the 64 distinct set-ups
        import oracle_lib
        fn = oracle_lib.load_oracle().lib.xo_esbr_sbr_frame_ps
        fn.restype = ctypes.c_int
        PF = ctypes.POINTER(ctypes.c_float)
        fn.argtypes = [PF] + [ctypes.c_void_p] * 6 + [PF, PF]
        lg, rg = out_l[:64].cpu().numpy(), out_r[:64].cpu().numpy()
        ok = True
        for i in range(len(recs)):
            a, b = np.zeros(2048, np.float32), np.zeros(2048, np.float32)
            so, po = new_state(), new_ps_state()
            fn(core_h[i].ctypes.data_as(PF), ctypes.byref(hs[i]), ctypes.byref(fs[i]), ctypes.byref(sds[i]), ctypes.byref(so),
               ctypes.byref(pfs[i]), ctypes.byref(po), a.ctypes.data_as(PF), b.ctypes.data_as(PF))
            ok = ok and np.array_equal(a.view(np.uint32), lg[i].view(np.uint32)) and np.array_equal(b.view(np.uint32), rg[i].view(np.uint32))
        ok = bool(ok)
    except Exception as e:
        ok = "unavailable: %r" % (e,)
    # the first frame of a stream comes with the header's reset flag (limiter tables are built); the steady state does not
    import esbr_structs
    sd.view(n, -1)[:, esbr_structs.EsbrSide.reset_flag.offset:esbr_structs.EsbrSide.reset_flag.offset + 2] = 0
    def timed(**kw):
        for i in range(max(warmup, 2) * len(lanes)):
            run_on(i % len(lanes), **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            run_on(i % len(lanes), **kw)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    ms = timed()
    refused = max([refused] + [float(l[6].cpu().numpy().astype(bool).mean()) for l in lanes])
    ab = n * (4096 + 2 * 8192 + 2 * st.shape[1] + 2 * pst.shape[1] + pf.shape[1] + hd.shape[1] + fr.shape[1] + sd.shape[1])
    # the same with every stream's QMF harmonic transposer tracked, as the reference runs it on each frame of such a stream
    # (docs/NOTEBOOK.md 5h; its output is only read by frames with harmonic SBR): two more launches per step
    from hbe_structs import state_from_tables
    hbs = [state_from_tables(h.freq_band_tbl_lo[:h.num_sf_bands[0] + 1], h.freq_band_tbl_hi[:h.num_sf_bands[1] + 1]) for h in hs]
    hb = tile(hbs)
    smax = max(int(x.synth_size) for x in hbs)   # the host knows its streams' bank sizes (xaac_hbe_state_reinit): the hint of the ABI
    hbq = [hb] + [hb.clone() for _ in lanes[1:]]
    _run_on = run_on
    run_on = lambda q, **kw: _run_on(q, hbe_state=hbq[q], hbe_max_synth_size=8 if smax <= 8 else 0)
    ms_h = timed()
    refused = max([refused] + [float(l[6].cpu().numpy().astype(bool).mean()) for l in lanes])
    with_transposer = {"value": round(n / ms_h * 1e3, 1), "ms_per_step": round(ms_h, 4), "launches": 7}
    return {"metric": "decoded audio frames/s (32-bit-ring QMF + float eSBR + float PS: the reference's default -esbr:1 path, HE-AACv2)",
            "value": round(n / ms * 1e3, 1), "unit": "frames/s", "steps": steps, "ms_per_step": round(ms, 4),
            "roofline_frac": round(ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "alg_bytes_per_step": int(ab), "dtype": "f32 / int64",
            "roofline_frac_on_ms_per_step": round(ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "refused_frac": refused, "bit_exact_vs_oracle": ok, "with_harmonic_transposer": with_transposer, "hip_streams": len(lanes),
            "timing": "wall clock over the steps (synchronised on both sides), steps dealt out over the HIP streams",
            "workload": "C4A: HE-AACv2 48 kHz, batch=%d streams/step, float core samples in: eSBR analysis -> float HF "
                        "generator + envelope adjuster -> float parametric stereo -> two eSBR synthesis banks (5 launches); "
                        "states carried from step to step" % n}


def usable_cores():
    """Host cores this process may really use: the scheduler affinity mask cut by the cgroup CPU quota (v2 cpu.max or
    v1 cfs_quota_us / cfs_period_us) when there is one.  Returns (workers, {"affinity":, "cgroup_quota":, "os_cpu_count":})."""
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                quota = None if q <= 0 else q / per
        except (OSError, ValueError):
            pass
    workers = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return workers, {"affinity": aff, "cgroup_quota": None if quota is None else round(quota, 2), "os_cpu_count": os.cpu_count()}


def timed_team(work, workers, seconds_budget, reset=None, max_passes=400):
    """Run work(t) on `workers` host threads at once (work = one ctypes call into a C loop: the GIL is released for its
    whole length) and time only the span between a common start line and the last thread's end: threads are created and
    every buffer is allocated BEFORE the clock starts; reset() (untimed) puts the inputs back between passes.  Returns
    (best seconds per pass, passes)."""
    best, spent, passes = 1e9, 0.0, 0
    while (spent < seconds_budget and passes < max_passes) or passes < 2:
        if reset is not None:
            reset()
        go = threading.Barrier(workers + 1)
        ends = [0.0] * workers

        def run(t):
            go.wait()
            work(t)
            ends[t] = time.perf_counter()

        th = [threading.Thread(target=run, args=(t,)) for t in range(workers)]
        [t.start() for t in th]
        go.wait()
        t0 = time.perf_counter()
        [t.join() for t in th]
        dt = max(ends) - t0
        best, spent, passes = min(best, dt), spent + dt, passes + 1
    return best, passes


def baseline_report(value, value_1, workers, info, kind, sample):
    """`cores` = the worker threads actually run; `scaling` = all-workers rate / one-worker rate, printed so that a reader
    sees whether the box really gave that many cores (a quota the cgroup files do not show, SMT siblings, memory
    bandwidth).  When the two disagree by more than 2 x the line says so instead of letting `cores` stand alone."""
    scaling = value / value_1 if value_1 else None
    out = {"value": round(value, 1), "unit": "frames/s", "cores": workers, "usable_cores": workers, "kind": kind,
           "value_1core": round(value_1, 1), "scaling": None if scaling is None else round(scaling, 2), "host": info,
           "sample": sample}
    if scaling is not None and scaling < 0.5 * workers:
        out["cores_note"] = ("%d worker threads ran (affinity mask / cgroup quota) but together they reached only %.1f x one "
                             "worker: the lease gives fewer physical cores than it lists (SMT siblings, a hidden quota) or the "
                             "pass is memory-bound; read `value` against `scaling`, not against `cores`" % (workers, scaling))
    return out


def cpu_baseline_sbr(workload, seconds_budget=10.0):
    """CPU baseline of the SBR workloads: the compiled reference's own ixheaacd_sbr_dec driven through
    oracle/ref_sbr_adapter.c (kind "reference") or, where oracle/_ref did not travel, the bit-exact restatement
    (kind "port"), on the committed reference-captured frames: every worker thread runs ONE C loop over its own shard
    of 1024-4096 channel-frames (arrays of the boundary structs); state and output buffers are allocated, and the state reset,
    outside the timed span."""
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
    workers, info = usable_cores()
    per_thread = max(1024, min(4096, 262144 // workers))   # channel-frames per worker per pass (23 KB of state + PCM each)
    pick = [recs[i % len(recs)] for i in range(per_thread)]
    blob = lambda key: b"".join(bytes(r[key]) for r in pick)
    hdr, frm, st0 = blob("header"), blob("frame"), blob("st0")
    pin = np.concatenate([r["pcm_in"] for r in pick]).astype(np.int16)
    psf, ps0 = (blob("ps_frame"), blob("ps0")) if hq else (None, None)
    fn.argtypes = ([ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p,
                    ctypes.c_void_p, ctypes.c_void_p] if hq else
                   [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p])
    st = [ctypes.create_string_buffer(st0, len(st0)) for _ in range(workers)]
    ps = [ctypes.create_string_buffer(ps0, len(ps0)) for _ in range(workers)] if hq else None
    out = [np.zeros(per_thread * (4096 if hq else 2048), np.int16) for _ in range(workers)]
    # ... and the core decoder's back end in front of it, as in the GPU step: one 1024-line IMDCT + overlap-add per
    # channel-frame (the reference's ixheaacd_imdct_process through oracle/ref_harness.c, or the restatement), on the
    # synthetic spectra the GPU leg uses.  The SBR calls keep their captured core samples: same work, two data sets
    P32, P16, PU8, p = oracle_lib.P32, oracle_lib.P16, oracle_lib.PU8, oracle_lib._p
    ref_imdct = kind == "reference" and hasattr(ref.lib, "ref_imdct_batch")
    orc = None if ref_imdct else oracle_lib.load_oracle()
    if ref_imdct:
        ref.lib.ref_imdct_batch.restype = None
        ref.lib.ref_imdct_batch.argtypes = [ctypes.c_int, P32, P32, P16, P16, PU8, PU8, P16]
    rng = np.random.default_rng(0xC0FFEE)
    spec0 = rng.integers(-(1 << 17), 1 << 17, (per_thread, 1024)).astype(np.int32)
    spec0[:, 640:] = 0
    ispec = [spec0.copy() for _ in range(workers)]
    iovl = [np.zeros((per_thread, 512), np.int32) for _ in range(workers)]
    iz16 = [np.zeros((2, per_thread), np.int16) for _ in range(workers)]
    iseq = np.zeros(per_thread, np.uint8)
    ishape = (np.arange(per_thread) % 2).astype(np.uint8)
    ipcm = [np.zeros((per_thread, 1024), np.int16) for _ in range(workers)]

    def reset():
        for t in range(workers):
            ctypes.memmove(st[t], st0, len(st0))
            if hq:
                ctypes.memmove(ps[t], ps0, len(ps0))
            if ref_imdct:
                ispec[t][:] = spec0      # the reference transforms its input in place

    def work(t):
        if ref_imdct:
            ref.lib.ref_imdct_batch(per_thread, p(ispec[t], P32), p(iovl[t], P32), p(iz16[t][0], P16), p(iz16[t][1], P16),
                                    p(iseq, PU8), p(ishape, PU8), p(ipcm[t], P16))
        else:
            orc.lib.xo_imdct_batch(per_thread, p(ispec[t], P32), p(iovl[t], P32), p(iz16[t][0], P16), p(iz16[t][1], P16),
                                   p(iseq, PU8), p(ishape, PU8), None, p(ipcm[t], P16), None, 0)
        if hq:
            fn(per_thread, hdr, frm, st[t], psf, ps[t], pin.ctypes.data, out[t].ctypes.data)
        else:
            fn(per_thread, hdr, frm, st[t], pin.ctypes.data, out[t].ctypes.data)

    timed_team(work, workers, 0.0, reset, max_passes=1)          # warm: pages touched, tables in cache
    t1, _ = timed_team(work, 1, 0.0, reset, max_passes=2)
    best, passes = timed_team(work, workers, seconds_budget, reset)
    per_frame = 1.0 if hq else 0.5     # C3 counts stereo frames: two channel calls each
    return baseline_report(workers * per_thread * per_frame / best, per_thread * per_frame / t1, workers, info, kind,
                           "%d ixheaacd_sbr_dec calls per worker per pass (one C loop per worker, buffers allocated and state "
                           "reset outside the timed span) on the committed reference-captured frames, each behind one "
                           "ixheaacd_imdct_process per channel-frame on the GPU leg's synthetic spectra (IMDCT + SBR chain, as "
                           "the GPU step), %d passes" % (per_thread, passes))


def cpu_baseline(seconds_budget=12.0, limiter=False):
    """Time the CPU path on a bounded sample of the same workload (every usable host core,
    one contiguous shard of channel-frames per worker).  limiter: each worker also runs the reference's
    ixheaacd_peak_limiter_process + round16 over its shard's frames (workload C2L)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    P32, P16, PU8 = oracle_lib.P32, oracle_lib.P16, oracle_lib.PU8
    ref = oracle_lib.load_reference()
    kind = "reference" if ref is not None and hasattr(ref.lib, "ref_imdct_batch") else "port"
    orc = oracle_lib.load_oracle()
    workers, info = usable_cores()
    per_thread = max(1024, min(4096, 262144 // workers))   # channel-frames per worker per pass
    rng = np.random.default_rng(0xC0FFEE)
    n = per_thread * workers
    base = rng.integers(-(1 << 17), 1 << 17, (per_thread, 1024)).astype(np.int32)
    base[:, 640:] = 0
    spec0 = np.tile(base, (workers, 1))
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

    def work(t):
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

    def reset():
        if kind == "reference":
            spec[:] = spec0           # the reference transforms its input in place

    timed_team(work, workers, 0.0, reset, max_passes=1)           # warm
    t1, _ = timed_team(work, 1, 0.0, reset, max_passes=2)
    best, passes = timed_team(work, workers, seconds_budget, reset, max_passes=2000)
    return baseline_report(n / CH / best, per_thread / CH / t1, workers, info, kind,
                           "%d stereo frames (%d channel-frames) per pass, %d passes, one %d-channel-frame shard per "
                           "worker thread (buffers allocated, input restored outside the timed span), same synthetic C2 input%s"
                           % (n // CH, n, passes, per_thread, " + limiter/round16 per frame" if limiter else ""))


METRIC = {
    "c2": "decoded audio frames/s (1024-spl IMDCT+overlap-add, AAC-LC stereo)",
    "c2l": "decoded audio frames/s (1024-spl IMDCT+overlap-add + peak limiter + PCM16, AAC-LC stereo)",
    "c3": "decoded audio frames/s (1024-spl IMDCT + 32/64-band QMF + low-power SBR, HE-AACv1 stereo)",
    "c4": "decoded audio frames/s (1024-spl IMDCT + complex QMF + HQ SBR + parametric stereo, HE-AACv2)",
}
WORKLOAD = {
    "c2": "C2: AAC-LC 48 kHz stereo, batch=8192 frames/step (16384 channel-frames), ONLY_LONG 1024-pt IMDCT + "
          "window/overlap-add + PCM16, %d stream sets cycled",
    "c2l": "C2L: C2 + the AAC-LC post stage, batch=8192 frames/step: IMDCT + window/overlap-add (WORD32 block + "
           "qshift_adj) -> peak limiter in place (5 ms look-ahead, attack / release smoothing) -> PCM16; every "
           + str(C2L_LOUD_EVERY) + "th stream decodes above full scale, %d stream sets cycled",
    "c3": "C3: HE-AACv1 48 kHz stereo, batch=8192 frames/step: IMDCT+OLA -> QMF-32 analysis -> LPP HF generation + "
          "envelope adjustment (side info cycled from reference-captured frames) -> QMF-64 synthesis, %d stream sets "
          "cycled",
    "c4": "C4: HE-AACv2 48 kHz, batch=8192 frames/step (one mono core channel each): IMDCT+OLA -> complex QMF-32 "
          "analysis -> LPP transposer + envelope adjustment -> parametric stereo -> two complex QMF-64 synthesis banks "
          "(SBR / PS side info cycled from reference-captured frames), %d stream sets cycled",
}
KERNELS = {
    "c2": "xaac_imdct_ola_kernel",
    "c2l": "imdct_ola + limiter_front + limiter_gain + limiter_apply (4 launches)",
    "c3": "imdct_ola + qmf_analysis + sbr_core_lp + qmf_synthesis (4 launches)",
    "c4": "imdct_ola + qmf_analysis_hq + sbr_core_hq (narrow rows; max_band_hint 48: no list launch) + ps + qmf_synthesis_pair (5 launches)",
}


def alg_bytes_per_step(w):
    return (C3_ALG_BYTES_PER_CH * CH if w == "c3" else C4_ALG_BYTES_PER_STREAM if w == "c4" else
            C2L_ALG_BYTES_PER_FRAME if w == "c2l" else ALG_BYTES_PER_FRAME) * FRAMES_PER_STEP


class Workload:
    """Inputs, per-step launch sequence and the self-check of one of the four workloads on one rank."""

    def __init__(self, w, torch, libxaac_amd, ctx, dev, stream, sets, seed, hip_streams=1):
        self.w, self.torch, self.x, self.ctx, self.dev, self.stream = w, torch, libxaac_amd, ctx, dev, stream
        c3, c4, c2l = w == "c3", w == "c4", w == "c2l"
        self.batches = (make_inputs_c3(torch, dev, sets, seed) if c3 else make_inputs_c4(torch, dev, sets, seed) if c4
                        else make_inputs(torch, dev, sets, seed, limiter=c2l))
        for b in self.batches:            # window shape alternates per frame; state follows
            b["ics"][:, 1] = (torch.arange(b["ics"].shape[0], device=dev) // (1 if c4 else CH) % 2).to(torch.uint8)
        n_units = FRAMES_PER_STEP * (1 if c4 else CH) if (c3 or c4) else FRAMES_PER_STEP
        self.status = torch.zeros(n_units, dtype=torch.int32, device=dev) if (c3 or c4 or c2l) else None
        self.imdct_status = torch.zeros(FRAMES_PER_STEP * (1 if c4 else CH), dtype=torch.int32, device=dev)
        self.ws = self._workspace(FRAMES_PER_STEP)
        # Steps dealt out over `hip_streams` HIP streams, step i on stream i % hip_streams, each with its own context, workspace
        # and status rows: the stream sets are independent batches (a host decoding many streams submits them exactly so),
        # and set j always meets stream j % hip_streams, so the frames of one set stay in order.  One kernel's tail (and the
        # persistent kernels' staggered start) then overlaps with the other stream's kernels instead of leaving CUs idle.
        assert hip_streams >= 1 and sets % hip_streams == 0, "every stream set has to stay on one HIP stream"
        self.lanes = [(ctx, stream, self.ws, self.status, self.imdct_status)]
        for st, cq in extra_lanes(torch, libxaac_amd, dev, hip_streams - 1):
            self.lanes.append((cq, st, self._workspace(FRAMES_PER_STEP), None if self.status is None else torch.zeros_like(self.status),
                               torch.zeros_like(self.imdct_status)))

    def _workspace(self, frames):
        t, ctx, w = self.torch, self.ctx, self.w
        nbytes = (ctx.sbr_lp_workspace_bytes(frames * CH) if w == "c3" else ctx.sbr_hq_workspace_bytes(frames, True)
                  if w == "c4" else ctx.peak_limiter_workspace_bytes(frames) if w == "c2l" else 0)
        return t.zeros(nbytes, dtype=t.uint8, device=self.dev) if nbytes else None

    def launch(self, b, frames_idx, k=None, ws=None, status=None, imdct_status=None, ctx=None):
        """one frame of the first k streams of batch b (k = None: all of them)"""
        ctx, x, w = (self.ctx if ctx is None else ctx), self.x, self.w
        ws = self.ws if ws is None else ws
        status = self.status if status is None else status
        imdct_status = self.imdct_status if imdct_status is None else imdct_status
        cut = (lambda t, per: t) if k is None else (lambda t, per: t[:k * per])
        if w == "c3":
            # core decoder back-end: planar PCM16 with the SBR hand-off rounding, then the low-power SBR chain
            ctx.imdct_process_batch(cut(b["spec"], CH), cut(b["ics"], CH), cut(b["overlap"], CH), cut(b["state"], CH), None,
                                    cut(b["core_pcm"], CH * 1024), None, ch_fac=1, pcm_mode=x.PCM_SBR,
                                    status=cut(imdct_status, CH))
            ctx.sbr_lp_process_batch(cut(b["core_pcm"], CH * 1024), cut(b["hdr"], CH),
                                     cut(b["frames"][frames_idx % len(b["frames"])], CH), cut(b["sbr_state"], CH),
                                     cut(b["pcm"], CH * 2048), ws, cut(status, CH), in_ch_fac=1, out_ch_fac=CH)
        elif w == "c4":
            # mono core back-end, then HQ SBR + parametric stereo -> L,R pairs
            ctx.imdct_process_batch(cut(b["spec"], 1), cut(b["ics"], 1), cut(b["overlap"], 1), cut(b["state"], 1), None,
                                    cut(b["core_pcm"], 1024), None, ch_fac=1, pcm_mode=x.PCM_SBR,
                                    status=cut(imdct_status, 1))
            fr, pfr = b["frames"][frames_idx % len(b["frames"])]
            ctx.sbr_hq_process_batch(cut(b["core_pcm"], 1024), cut(b["hdr"], 1), cut(fr, 1), cut(b["sbr_state"], 1),
                                     cut(b["pcm"], 4096), ws, cut(pfr, 1), cut(b["ps_state"], 1), cut(status, 1),
                                     max_band_hint=b.get("max_band_hint", 0))
        elif w == "c2l":
            # AAC-LC tail as api.c runs it: WORD32 block + qshift_adj -> limiter in place -> interleaved PCM16.  The
            # block stays planar between the two (16-byte stores from the IMDCT; the limiter interleaves on the way out)
            ctx.imdct_process_batch(cut(b["spec"], CH), cut(b["ics"], CH), cut(b["overlap"], CH), cut(b["state"], CH),
                                    cut(b["out32"], CH * 1024), None, cut(b["qshift"], CH), ch_fac=1,
                                    status=cut(imdct_status, CH))
            ctx.peak_limiter_process_batch(cut(b["out32"], CH * 1024), cut(b["qshift"], CH), cut(b["lim_state"], 1), CH, ws,
                                           pcm16=cut(b["pcm"], CH * 1024), planar=True, status=cut(status, 1))
        else:
            ctx.imdct_process_batch(cut(b["spec"], CH), cut(b["ics"], CH), cut(b["overlap"], CH), cut(b["state"], CH), None,
                                    cut(b["pcm"], CH * 1024), None, ch_fac=CH, pcm_mode=x.PCM_LC,
                                    status=cut(imdct_status, CH))

    def step(self, i, ev=None):
        b = self.batches[i % len(self.batches)]
        ctx, stream, ws, status, imdct_status = self.lanes[i % len(self.lanes)]
        if ev is not None:
            ev[0].record(stream)
        self.launch(b, i // len(self.batches), ws=ws, status=status, imdct_status=imdct_status, ctx=ctx)
        if ev is not None:
            ev[1].record(stream)

    def run(self, steps, warmup, barrier, prewarm=0):
        """(wall seconds of the K steps, milliseconds of GPU time per step from HIP events on the launch streams).  One HIP
        stream: the mean of the steps' own event pairs.  Several: the steps of different streams overlap, so a step's own pair
        says little; the span from the first stream's start event to the last stream's end event, over K.
        prewarm: untimed steps in front of the W counted warm-up steps (config.prewarm_steps).  A process that has just built
        its inputs meets a device that is not in its steady state yet -- measured, profiles/r06_a_clock_probe.txt: 20 steps
        behind 5 warm-up steps take 0.525-0.540 ms each, the same 20 steps behind 60 take 0.488-0.509, like the steps of a
        100- or 400-step run (0.490-0.504); a step is 0.5 ms, so W = 5 is 2.5 ms of work after seconds of host-side set-up."""
        torch = self.torch
        for i in range(prewarm):
            self.step(i)
        for i in range(prewarm, prewarm + warmup):
            self.step(i)
        warmup += prewarm
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        many = len(self.lanes) > 1
        span = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in self.lanes]
        barrier()
        t0 = time.perf_counter()
        for q, lane in enumerate(self.lanes):
            span[q][0].record(lane[1])
        for i in range(steps):
            self.step(warmup + i, None if many else events[i])
        for q, lane in enumerate(self.lanes):
            span[q][1].record(lane[1])
        barrier()
        elapsed = time.perf_counter() - t0
        if many:
            kern_ms = max(a[0].elapsed_time(b[1]) for a in span for b in span) / max(1, steps)
        else:
            kern_ms = float(np.mean([a.elapsed_time(b) for a, b in events])) if events else float("nan")
        self.next_step = warmup + steps
        return elapsed, kern_ms

    def refused(self):
        """fraction of the last step's units the kernels refused (status != 0): must be zero for the timed work to
        be the whole work"""
        bad = 0.0
        for _, _, _, status, imdct_status in self.lanes:
            if status is not None:
                bad = max(bad, float((status != 0).float().mean().item()))
            bad = max(bad, float((imdct_status != 0).float().mean().item()))
        return bad

    def verify(self, k=256):
        """Decode one more frame of the first k streams of the next batch twice from the same device state: by
        the oracle chain on the host and by the GPU path on the slice; PCM, carried state and status must agree
        word for word."""
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        torch, w = self.torch, self.w
        orc = oracle_lib.load_oracle()
        P16 = ctypes.POINTER(ctypes.c_int16)
        i = self.next_step
        b = self.batches[i % len(self.batches)]
        fidx = i // len(self.batches)
        cf = 1 if w == "c4" else CH
        h = {n_: b[n_][:k * cf].cpu().numpy() for n_ in ("spec", "ics", "overlap", "state")}
        torch.cuda.synchronize()
        if w == "c2":
            want = orc.imdct_batch(h["spec"], h["ics"], h["overlap"], h["state"], ch_fac=CH)
            self.launch(b, fidx, k=k)
            torch.cuda.synchronize()
            return bool(np.array_equal(b["pcm"][:k * CH * 1024].cpu().numpy().reshape(k * CH, 1024), want["pcm16"]) and
                        np.array_equal(b["overlap"][:k * CH].cpu().numpy(), want["overlap"]))
        if w == "c2l":
            import limiter_cases as lc
            core = orc.imdct_batch(h["spec"], h["ics"], h["overlap"], h["state"], ch_fac=CH)   # interleaved WORD32 block
            xs = np.ascontiguousarray(core["out32"].reshape(-1))
            q = np.ascontiguousarray(core["qshift_adj"])
            st_raw = np.ascontiguousarray(b["lim_state"][:k].cpu().numpy())
            st = (lc.LimiterState * k).from_buffer_copy(st_raw.tobytes())
            pcm = np.zeros(k * CH * 1024, np.int16)
            lc.bind(orc.lib, "xo")[2](k, 1024, CH, xs.ctypes.data_as(lc.P32), 1024 * CH, q.ctypes.data_as(lc.P8), st,
                                       pcm.ctypes.data_as(lc.P16))
            ws = self._workspace(k)
            self.launch(b, fidx, k=k, ws=ws)
            torch.cuda.synchronize()
            got_st = (lc.LimiterState * k).from_buffer_copy(b["lim_state"][:k].cpu().numpy().tobytes())
            return bool(np.array_equal(b["pcm"][:k * CH * 1024].cpu().numpy(), pcm) and
                        all(lc.state_view(got_st[j]) == lc.state_view(st[j]) for j in range(k)))
        import sbr_capture as cap
        core = orc.imdct_batch(h["spec"], h["ics"], h["overlap"], h["state"], pcm_mode=1)
        n = k * cf
        hdr = b["hdr"][:n].cpu().numpy()
        st = np.ascontiguousarray(b["sbr_state"][:n].cpu().numpy())
        frames = b["frames"][fidx % len(b["frames"])]
        if w == "c4":
            fr, pfr = frames[0][:n].cpu().numpy(), frames[1][:n].cpu().numpy()
            ps = np.ascontiguousarray(b["ps_state"][:n].cpu().numpy())
        else:
            fr = frames[:n].cpu().numpy()
        out = np.zeros(n * 2048 * (2 if w == "c4" else 1), np.int16)
        rcs = np.zeros(n, np.int32)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        for j in range(n):
            pin = np.ascontiguousarray(core["pcm16"][j])
            if w == "c4":
                rcs[j] = orc.lib.xo_sbr_dec_hq(vp(hdr[j]), vp(fr[j]), vp(st[j]), vp(pfr[j]), vp(ps[j]), pin.ctypes.data_as(P16),
                                               1, out[4096 * j:].ctypes.data_as(P16), 2)
            else:
                rcs[j] = orc.lib.xo_sbr_dec_lp(vp(hdr[j]), vp(fr[j]), vp(st[j]), pin.ctypes.data_as(P16), 1,
                                               out[(j // CH) * 2048 * CH + (j % CH):].ctypes.data_as(P16), CH)
        ws = self._workspace(k)
        self.launch(b, fidx, k=k, ws=ws)
        torch.cuda.synchronize()
        ok = np.array_equal(self.status[:n].cpu().numpy(), rcs)
        good = rcs == 0
        per = 4096 if w == "c4" else 2048
        g_pcm = b["pcm"][:n * per].cpu().numpy()
        if w == "c4":
            ok = ok and np.array_equal(g_pcm.reshape(n, per)[good], out.reshape(n, per)[good])
            ok = ok and np.array_equal(b["ps_state"][:n].cpu().numpy()[good], ps[good])
        else:
            unint = lambda a: a.reshape(k, 2048, CH).transpose(0, 2, 1).reshape(n, 2048)
            ok = ok and np.array_equal(unint(g_pcm)[good], unint(out)[good])
        ok = ok and np.array_equal(b["sbr_state"][:n].cpu().numpy()[good], st[good])
        ok = ok and np.array_equal(b["overlap"][:n].cpu().numpy(), core["overlap"])
        return bool(ok)


def need_gpus(torch, n):
    """the product has no CPU path: say what is missing instead of failing somewhere inside a launch"""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        sys.exit("bench.py needs %d GPU%s on this node (found %d); the product has no CPU path" % (n, "s" if n > 1 else "", have))


def self_launch(n, launch_check):
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves, the way the driver's own command line
    does (one process per GPU, rendezvous on 127.0.0.1), and hand back the launcher's exit code.  The device count is
    checked first so that a box without N GPUs says so."""
    import socket
    import subprocess
    if not launch_check:
        import torch
        need_gpus(torch, 1 if "--share-device" in sys.argv else n)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def launch_check(torch, xdist, rank, world):
    """--launch-check: everything a multi-rank run does around its timed region except the decode, on host tensors over
    gloo: rendezvous, barrier, max-over-ranks, per-rank report and the gather of a per-rank PCM-shaped pattern into rank 0."""
    dist = xdist.init("gloo")
    dev = torch.device("cpu")
    barrier = dist.barrier if dist is not None else (lambda: None)
    barrier()
    elapsed = xdist.max_over_ranks(dist, 1.0 + rank, dev)
    pcm = (torch.arange(64 * 4096, dtype=torch.int32).view(64, 4096) * (rank + 3) % 65521 - 32760).to(torch.int16)
    per_rank, gather = (None, None) if dist is None else xdist.post_run_report(dist, pcm, 1000.0 + rank, dev, barrier)
    if rank == 0:
        print(json.dumps({"launch_check": True, "metric": None, "value": None, "n_gpus": world, "max_over_ranks_s": elapsed,
                          "per_rank_frames_per_s": per_rank, "gather": gather,
                          "note": "launcher / rendezvous / gather check on host tensors (gloo); not a measurement"}))
    if dist is not None:
        assert gather["ok"] is not False
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--prewarm", type=int, default=PREWARM_STEPS,
                    help="untimed steps in front of the W counted warm-up steps, so that a short run (the driver's --steps 20 "
                         "--warmup 5) is timed on a device in the same state as a long one (profiles/r06_a_clock_probe.txt)")
    ap.add_argument("--sets", type=int, default=4, help="independent 8192-stream batches cycled per rank")
    ap.add_argument("--hip-streams", type=int, default=2,
                    help="HIP streams per rank the steps are dealt out over (step i on stream i %% N; must divide --sets): "
                         "neighbouring steps decode different stream sets, so they share nothing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short C2 / C3 runs of the N=1 report")
    ap.add_argument("--workload", choices=["c2", "c2l", "c3", "c4"], default="c4",
                    help="c4 (default): HE-AACv2, mono IMDCT + HQ-SBR + parametric stereo -- the configuration the "
                         "metric is quoted on, and C5 with --gpus N; c2: AAC-LC IMDCT+OLA (BASELINE configs[1]); c2l: c2 "
                         "+ peak limiter + PCM16 (the AAC-LC post stage); c3: HE-AACv1 stereo, IMDCT + LP-SBR")
    ap.add_argument("--launch-check", action="store_true",
                    help="no decode, no GPU: start the N ranks exactly as a real run does, rendezvous over gloo and run the "
                         "post-run report (per-rank rates + the PCM gather) on host tensors; prints a line that is NOT a "
                         "measurement (tests/test_dist_cpu.py)")
    ap.add_argument("--share-device", action="store_true",
                    help="a box with fewer GPUs than ranks: rank r decodes on device r mod the node's count and the ranks meet "
                         "over gloo on host tensors instead of RCCL (tests/test_dist_gpu.py: the N > 1 path of the HIP product on "
                         "a one-GPU box; the line says shared_device and is not a scaling measurement)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, args.launch_check))

    import torch
    import libxaac_amd
    # torch sizes its CPU pool by the CPUs the machine lists; on a box that lists 256 and grants 16 the pool's spinning threads
    # would eat the cgroup quota the host-side legs (cpu_baseline, end_to_end) are measured in
    torch.set_num_threads(max(1, min(torch.get_num_threads(), usable_cores()[0])))

    from libxaac_amd import dist as xdist
    rank, local_rank, world = xdist.env_rank()
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher's WORLD_SIZE is %d" % (args.gpus, world))
    if args.launch_check:
        return launch_check(torch, xdist, rank, world)
    shared = bool(args.share_device and world > 1)
    if shared:
        need_gpus(torch, 1)
        local_rank = local_rank % torch.cuda.device_count()
    else:
        need_gpus(torch, max(args.gpus, local_rank + 1))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = xdist.init("gloo" if shared else "nccl")       # RCCL; None at world 1
    cdev = torch.device("cpu") if shared else dev         # where the collectives' tensors live
    if dist is not None:
        assert dist.get_world_size() == args.gpus and dist.get_rank() == rank

    stream = torch.cuda.Stream(device=dev)      # kernels AND timing events go on this one stream
    torch.cuda.set_stream(stream)
    ctx = libxaac_amd.XaacContext(local_rank, stream.cuda_stream)

    def barrier():
        if dist is not None:
            if shared:
                torch.cuda.synchronize()
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])   # the rank's own GPU, said explicitly
        torch.cuda.synchronize()

    w = args.workload
    job = Workload(w, torch, libxaac_amd, ctx, dev, stream, args.sets, rank, hip_streams=args.hip_streams)
    own_elapsed, kern_ms = job.run(args.steps, args.warmup, barrier, prewarm=args.prewarm)
    elapsed = xdist.max_over_ranks(dist, own_elapsed, cdev)
    refused = job.refused()
    max_band_hint = job.batches[0].get("max_band_hint", 0) if w == "c4" else None
    try:
        checked = job.verify()
    except Exception as e:  # the checker (oracle/) is test infrastructure: its absence must not hide the measurement
        checked = "unavailable: %r" % (e,)

    # the ranks' own rates (load balance of a SCALE run) and the final interleaved gather over RCCL/xGMI, once
    per_rank, gather = None, None
    if dist is not None:
        pcm = job.batches[0]["pcm"].view(FRAMES_PER_STEP, -1)
        per_rank, gather = xdist.post_run_report(dist, pcm.cpu() if shared else pcm, FRAMES_PER_STEP * args.steps / own_elapsed, cdev, barrier)
        assert gather["ok"] is not False, "the gathered PCM differs from the ranks' shards"

    secondary = None
    if world == 1 and not args.no_secondary and w == "c4":
        secondary = {}
        try:   # what a caller with smaller batches gets: the same launches per step on the first k streams
            sweep = {}
            for k in (256, 1024, 4096):
                ws_k = [job._workspace(k) for _ in job.lanes]

                def small(i, k=k, ws_k=ws_k):
                    q = i % len(job.lanes)
                    ctx_q, _, _, st_q, ist_q = job.lanes[q]
                    job.launch(job.batches[i % len(job.batches)], i // len(job.batches), k=k, ws=ws_k[q], status=st_q,
                               imdct_status=ist_q, ctx=ctx_q)

                for i in range(4):
                    small(i)
                barrier()
                t0 = time.perf_counter()
                for i in range(40):
                    small(i)
                barrier()
                dt = (time.perf_counter() - t0) / 40
                sweep[str(k)] = {"ms_per_step": round(dt * 1e3, 4), "frames_per_s": round(k / dt, 1)}
            sweep["note"] = ("C4 chain on the first k streams of the batches (five launches per step, no HIP graph), steps dealt out "
                             "over the run's %d HIP streams like the headline's: one stream-frame takes about 0.11 ms through the "
                             "six kernels whatever the batch (one wave per stream in the two long ones), so below a few thousand "
                             "streams a step is that latency and most of the chip idles unless independent batches overlap"
                             % len(job.lanes))
            secondary["c4_batch_sweep"] = sweep
        except Exception as e:
            secondary["c4_batch_sweep"] = {"error": repr(e)}
        del job.batches, job.ws
        torch.cuda.empty_cache()
        for w2 in ("c2", "c3"):
            j2 = Workload(w2, torch, libxaac_amd, ctx, dev, stream, args.sets, rank, hip_streams=args.hip_streams)
            e2, k2 = j2.run(max(20, args.steps // 2), max(4, args.warmup // 2), barrier, prewarm=args.prewarm)
            steps2 = max(20, args.steps // 2)
            ab = alg_bytes_per_step(w2)
            try:
                ok2 = j2.verify()
            except Exception as e:
                ok2 = "unavailable: %r" % (e,)
            secondary[w2] = {"metric": METRIC[w2], "value": round(FRAMES_PER_STEP * steps2 / e2, 1), "unit": "frames/s",
                             "steps": steps2, "ms_per_step": round(e2 / steps2 * 1e3, 4), "kernel_ms": round(k2, 5),
                             "roofline_frac": round(ab / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "roofline_frac_on_ms_per_step": round(ab / (e2 / steps2) / 1e9 / HBM_PEAK_GBS, 4),
                             "alg_bytes_per_step": ab, "refused_frac": j2.refused(), "bit_exact_vs_oracle": ok2,
                             "workload": WORKLOAD[w2] % args.sets}
            del j2
            torch.cuda.empty_cache()
        try:
            secondary["c4_esbr"] = secondary_esbr(torch, libxaac_amd, ctx, dev, max(10, args.steps // 5), 2, hip_streams=args.hip_streams)
        except Exception as e:  # never lose the headline line over the extra entry
            secondary["c4_esbr"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        try:
            secondary["f4_transforms"] = secondary_f4(torch, libxaac_amd, ctx, dev, hip_streams=args.hip_streams)
        except Exception as e:
            secondary["f4_transforms"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        try:
            secondary["round6_rows"] = secondary_round6(torch, libxaac_amd, ctx, dev)
        except Exception as e:
            secondary["round6_rows"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        try:
            secondary["end_to_end"] = end_to_end_in_its_own_process()
        except Exception as e:
            secondary["end_to_end"] = {"error": repr(e)}
        torch.cuda.empty_cache()

    if rank == 0:
        frames = FRAMES_PER_STEP * args.steps * world
        alg_bytes = alg_bytes_per_step(w)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        pmc = measured_traffic(w)
        out = {
            "metric": METRIC[w],
            "value": round(frames / elapsed, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32 (+ f32/f64 gain smoothing)" if w == "c2l" else "int32",
            "data": "synthetic",
            "config": {"workload": WORKLOAD[w] % args.sets,
                       "frames_per_step": FRAMES_PER_STEP, "channels": 1 if w == "c4" else CH, "launch": ctx.last_launch(),
                       "hip_streams": args.hip_streams, "prewarm_steps": args.prewarm,
                       "max_band_hint": max_band_hint,
                       "prewarm_note": "untimed steps in front of the W counted warm-up steps: the timed K steps of a short run "
                                       "then meet the device in the state a long run times it in",
                       "hip_streams_note": "step i is launched on HIP stream i % hip_streams (its own context and workspace); "
                                           "a stream set always meets the same stream, so its frames stay in order",
                       "sharding": "streams split across ranks (%d per rank), no data-path collective" % FRAMES_PER_STEP},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": pmc.get("bytes_per_step"), "traffic_source": pmc.get("source"),
                         "kernel": KERNELS[w], "kernel_ms": round(kern_ms, 5), "alg_bytes_per_launch": alg_bytes,
                         "valu": valu_roofline(w, kern_ms)},
            "refused_frac": refused,
            "bit_exact_vs_oracle": checked,
        }
        if per_rank is not None:
            out["per_rank_frames_per_s"] = per_rank
            out["gather"] = gather
        if shared:
            out["shared_device"] = ("%d ranks on %d device(s), rendezvous / max-over-ranks / PCM gather over gloo on host tensors: the N > 1 "
                                    "path of the product on a box with fewer GPUs than ranks -- NOT a scaling measurement"
                                    % (world, torch.cuda.device_count()))
        if secondary is not None:
            out["secondary"] = secondary
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_sbr(w) if w in ("c3", "c4") else cpu_baseline(limiter=(w == "c2l"))
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
