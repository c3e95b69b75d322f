#!/usr/bin/env python3
"""End-to-end throughput of the batched multi-stream host (oracle/_ref/xaacdec_batch): parse (the reference's own
decoder instances on the host cores) + conversion + PCIe both ways + GPU, next to the same streams decoded by as many
plain reference decoders (oracle/_ref/xaacdec) run in parallel on the same cores.  A 20 s HE-AACv2 stream is made on
the spot with the reference encoder (oracle/_ref/xaacenc).  Prints one JSON line.  Run on the GPU box:
    python tools/bench_batch_host.py [instances] [groups]"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_test_streams as mts  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    groups = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    seconds = 20.0
    tmp = tempfile.mkdtemp(prefix="xaac_batch_")
    t = np.arange(int(48000 * seconds)) / 48000.0
    rng = np.random.default_rng(7)
    tone = sum((0.3 / k) * np.sin(2 * np.pi * 440.0 * k * t + k) for k in range(1, 24)) * (0.6 + 0.4 * np.sin(2 * np.pi * 0.9 * t))
    nz = rng.standard_normal(len(t)) * (0.03 + 0.15 * (np.floor(t * 2.5) % 2))
    x = np.stack([0.5 * tone + nz, 0.5 * tone + 0.6 * nz[::-1]], 1)
    wav, aac = os.path.join(tmp, "in.wav"), os.path.join(tmp, "in.aac")
    mts.write_wav(wav, x)
    subprocess.run([os.path.join(REF, "xaacenc"), "-ifile:" + wav, "-ofile:" + aac, "-aot:29", "-br:32000", "-adts:1"],
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    frames_per_stream = int(seconds * 48000 / 2048)
    per = n // groups
    args = [os.path.join(REF, "xaacdec_batch"), "-esbr:0", "--"] + ["%d:%s:%s/g%d" % (per, aac, tmp, k) for k in range(groups)]
    t0 = time.perf_counter()
    p = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    wall = time.perf_counter() - t0
    summary = json.loads(p.stdout.decode().strip().splitlines()[-1])
    frames = summary["calls"]["sbr_ps"] + summary["calls"]["sbr_hq"]
    # the same number of plain reference decoders, all at once, on the same cores
    t0 = time.perf_counter()
    procs = [subprocess.Popen([os.path.join(REF, "xaacdec"), "-ifile:" + aac, "-ofile:%s/r%d.wav" % (tmp, i), "-esbr:0"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i in range(per * groups)]
    for q in procs:
        q.wait()
    wall_ref = time.perf_counter() - t0
    same = open("%s/g0.0.wav" % tmp, "rb").read() == open("%s/r0.wav" % tmp, "rb").read()
    # the batched host again with the reference's DEFAULT flags: the SBR seam goes through the float eSBR chain
    args_d = [os.path.join(REF, "xaacdec_batch"), "--"] + ["%d:%s:%s/e%d" % (per, aac, tmp, k) for k in range(groups)]
    pd = subprocess.run(args_d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    sd = json.loads(pd.stdout.decode().strip().splitlines()[-1])
    frames_d = sd["calls"]["esbr_ps"] + sd["calls"]["esbr"]
    # ... and with the reference's default flags (-esbr:1: its float eSBR path, the harmonic transposer running idle)
    t0 = time.perf_counter()
    procs = [subprocess.Popen([os.path.join(REF, "xaacdec"), "-ifile:" + aac, "-ofile:%s/d%d.wav" % (tmp, i)],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i in range(per * groups)]
    for q in procs:
        q.wait()
    wall_def = time.perf_counter() - t0
    print(json.dumps({"what": "HE-AACv2 48 kHz, %d decoder instances in %d groups, %.0f s stream each: reference parser "
                              "instances + conversion + PCIe + GPU back-end, state shipped both ways per call" % (per * groups, groups, seconds),
                      "frames": frames, "frames_expected_per_stream": frames_per_stream,
                      "end_to_end_frames_per_s": round(frames / summary["seconds"], 1), "seconds": summary["seconds"],
                      "wall_incl_fork": round(wall, 3), "pinned": summary["pinned"], "batches": summary["batches"],
                      "cpu_only_reference_frames_per_s": round(frames / wall_ref, 1),
                      "cpu_only_reference_default_flags_frames_per_s": round(frames / wall_def, 1),
                      "default_flags_end_to_end_frames_per_s": round(frames_d / sd["seconds"], 1), "default_flags_batches": sd["batches"],
                      "default_flags_output_identical": open("%s/e0.0.wav" % tmp, "rb").read() == open("%s/d0.wav" % tmp, "rb").read(),
                      "cores": os.cpu_count(),
                      "output_identical": same}))


if __name__ == "__main__":
    main()
