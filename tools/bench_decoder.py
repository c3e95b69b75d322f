#!/usr/bin/env python3
"""End-to-end rate of the repo's own decoder (libxaac_amd/decoder.py): N copies of a committed ADTS stream decoded in lock
step -- host parser threads -> pinned staging -> H2D -> GPU entry points on device-resident state -> D2H PCM -- and the host
parser alone beside it (the ceiling of any host that parses on this machine's CPU).  Prints one JSON object."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parser_only(decoder, data, n, threads, frames_per_call=4):
    """the host parser alone, the way decode_streams drives it: xaac_parse_batch_start / _wait with frames_per_call frames of
    every stream per call (plain host arrays; nothing is copied anywhere)"""
    bp = decoder.BatchParser([data] * n, threads=threads)
    from libxaac_amd import PS_FRAME_BYTES, SBR_FRAME_BYTES, SBR_HEADER_BYTES
    nc, T = n * bp.n_ch, int(frames_per_call)
    spec, ics = np.zeros((T, nc, 1024), np.int32), np.zeros((T, nc, 2), np.uint8)
    hdr, frm = np.zeros((T, nc, SBR_HEADER_BYTES), np.uint8), np.zeros((T, nc, SBR_FRAME_BYTES), np.uint8)
    psf, flags = np.zeros((T, n, PS_FRAME_BYTES), np.uint8), np.zeros((T, n, 8), np.int32)
    status, pitch = np.zeros((T, n), np.int32), np.zeros((T, n), np.int32)
    sbr = bp.sbr
    t0 = time.perf_counter()
    while True:
        bp.start_step(spec, ics, hdr if sbr else None, frm if sbr else None, psf if sbr and bp.n_ch == 1 else None,
                      flags if sbr else None, status=status, reset_pitch=pitch, frames=T)
        bp.wait_step(check=False)
        live = False
        for t in range(T):
            live = bool(bp.finish_step(None, status[t]).any()) or live
        if not live:
            break
    dt = time.perf_counter() - t0
    frames = int(bp.frames.sum())
    bp.close()
    return frames / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stream", default="mix_aot29_32k")
    ap.add_argument("--copies", type=int, default=2048)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--overlap", type=int, default=1)
    a = ap.parse_args()
    from libxaac_amd import decoder
    data = open(os.path.join(ROOT, "tests", "golden", "streams", a.stream + ".aac"), "rb").read()
    decoder.decode_streams([data] * 8)                           # warm up (library load, first launches)
    timing = {}
    t0 = time.perf_counter()
    decoder.decode_streams([data] * a.copies, threads=a.threads, keep_pcm=False, timing=timing, overlap=bool(a.overlap))
    wall = time.perf_counter() - t0
    out = {"stream": a.stream, "copies": a.copies, "frames": timing["frames"], "wall_s": round(wall, 4),
           "steps_s": round(timing["steps_s"], 4), "parse_s": round(timing["parse_s"], 4), "gpu_s": round(timing["gpu_s"], 4),
           "end_to_end_frames_per_s": round(timing["frames"] / timing["steps_s"], 1),
           "parser_only_frames_per_s": round(parser_only(decoder, data, a.copies, a.threads), 1),
           "host_threads": a.threads or os.cpu_count(), "overlap": bool(a.overlap)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
