#!/usr/bin/env python3
"""Phase timers of the SBR core kernel: builds the product library with -DXS_PROFILE (the XS_T hooks of
sbr_core.h accumulate lane-0 cycle counts per phase), runs the C3 bench inputs through it a few times and
prints cycles per channel-frame for every phase.  Developer tool; run on the GPU box."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {1: "block floating point", 2: "hf generator (total)", 3: "env: init/sineflags/adj_e", 4: "env: energies",
         5: "env: gain meta (lane 0)", 6: "env: subband gains", 7: "env: noise limiting", 8: "env: alias groups+reduction (LP) / limiter bands, regular-frame test (HQ)",
         9: "env: erg->amplitude (LP) / band maps (HQ)", 10: "env: apply (adapt_noise_gain)", 11: "env: final adjust", 12: "hf: setup/clears",
         13: "hf: lpc coefficients (+covariance in LP)", 14: "hf: degree alias (LP) / patches (HQ)", 15: "tail (state, lpc save)", 16: "lim: avggain", 17: "lim: limit loop", 18: "lim: accumulate",
         19: "lim: boost div", 20: "lim: scale loop", 21: "apply: startup/equalize/tones", 22: "apply: slot loop",
         23: "alias: groups (LP) / pass set-up (HQ)", 24: "copy-in (global -> LDS)", 25: "copy-out (LDS -> global)",
         26: "side-info check, narrow-row check, rescale overlap", 27: "bfp: headroom scans", 28: "env: erg->amplitude (pairs)",
         29: "apply: equalize filt buf", 30: "apply: smoothed slots", 19: "lim: boost div", 31: "hf: covariance sums"}


PS_NAMES = {1: "sanitize, init_ps_scale", 2: "P1 hybrid analysis", 3: "P2 envelope walk", 4: "P3 band powers, inputs",
            5: "P4 transient detector", 6: "P5 all-pass chains", 7: "P6 hybrid rotation", 8: "P7 delays, rotation, rows out"}


def build_prof(src, out):
    """the product library with the XS_T / XP_T timer hooks compiled in (its own object directory)"""
    subprocess.check_call(["make", "-s", "-C", src, "OUT=" + out, "OBJD=" + os.path.join(src, "build_prof"),
                           "EXTRA=-DXS_PROFILE"])


def main_ps():
    """same for the parametric-stereo kernel on the C4 bench inputs: python tools/prof_sbr_core.py ps"""
    import torch
    import libxaac_amd
    src = os.path.join(ROOT, "libxaac_amd", "csrc")
    out = os.path.join(ROOT, "libxaac_amd", "libxaac_amd_prof.so")
    build_prof(src, out)
    libxaac_amd.library_path = lambda: out
    import bench
    dev = torch.device("cuda:0")
    n = bench.FRAMES_PER_STEP
    b = bench.make_inputs_c4(torch, dev, 1, 0)[0]
    ctx = libxaac_amd.XaacContext(0, None)
    ws = torch.zeros(ctx.sbr_hq_workspace_bytes(n, True), dtype=torch.uint8, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    steps = 4
    for i in range(steps):
        ctx.imdct_process_batch(b["spec"], b["ics"], b["overlap"], b["state"], None, b["core_pcm"], None, ch_fac=1,
                                pcm_mode=libxaac_amd.PCM_SBR)
        fr, pfr = b["frames"][i % 4]
        ctx.sbr_hq_process_batch(b["core_pcm"], b["hdr"], fr, b["sbr_state"], b["pcm"], ws, pfr, b["ps_state"], status)
    ctx.sync()
    raw = status.cpu().numpy()[:128].view(np.uint64).astype(np.float64) / (steps * n)
    core, ps = raw[:32], raw[32:48]
    print("HQ core kernel:")
    for i in range(1, 32):
        if core[i]:
            print("%2d %-34s %9.0f cycles/channel-frame %5.1f%%" % (i, NAMES.get(i, "?"), core[i], 100 * core[i] / core.sum()))
    print("   total %.0f cycles" % core.sum())
    print("PS kernel:")
    for i in range(1, 9):
        print("%2d %-34s %9.0f cycles/stream-frame %5.1f%%" % (i, PS_NAMES[i], ps[i], 100 * ps[i] / ps.sum()))
    print("   total %.0f cycles" % ps.sum())


def main():
    import torch
    import libxaac_amd
    src = os.path.join(ROOT, "libxaac_amd", "csrc")
    out = os.path.join(ROOT, "libxaac_amd", "libxaac_amd_prof.so")
    build_prof(src, out)
    libxaac_amd.library_path = lambda: out
    import bench
    dev = torch.device("cuda:0")
    n = bench.FRAMES_PER_STEP * bench.CH
    b = bench.make_inputs_c3(torch, dev, 1, 0)[0]
    ctx = libxaac_amd.XaacContext(0, None)
    ws = torch.zeros(ctx.sbr_lp_workspace_bytes(n), dtype=torch.uint8, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    steps = 4
    for i in range(steps):
        ctx.imdct_process_batch(b["spec"], b["ics"], b["overlap"], b["state"], None, b["core_pcm"], None, ch_fac=1,
                                pcm_mode=libxaac_amd.PCM_SBR)
        ctx.sbr_lp_process_batch(b["core_pcm"], b["hdr"], b["frames"][i % 4], b["sbr_state"], b["pcm"], ws, status,
                                 in_ch_fac=1, out_ch_fac=2)
    ctx.sync()
    acc = status.cpu().numpy()[:64].view(np.uint64).astype(np.float64) / (steps * n)
    tot = acc.sum()
    for i in range(1, 24):
        print("%2d %-34s %9.0f cycles/channel-frame %5.1f%%" % (i, NAMES[i], acc[i], 100 * acc[i] / tot))
    print("   total %.0f cycles" % tot)
    # the low-power synthesis bank's own timers (sbr_qmf_kernel.hip: XQ_TIME, counters 64 ..; one wave = a pair of channels)
    syn = status.cpu().numpy()[:160].view(np.uint64).astype(np.float64)[64:70] / (steps * n / 2)
    names = ["wait for the last pair's stores", "rows in (global -> LDS tile)", "rescale + inverse modulation (lane = slot)",
             "ring samples + history into LDS", "window-add, PCM out", "ring state out"]
    print("LP synthesis bank (cycles per pair of channel-frames):")
    for i in range(6):
        print("   %-44s %9.0f %5.1f%%" % (names[i], syn[i], 100 * syn[i] / max(1.0, syn.sum())))


if __name__ == "__main__":
    main_ps() if len(sys.argv) > 1 and sys.argv[1] == "ps" else main()
