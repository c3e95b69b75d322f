#!/usr/bin/env python3
"""Developer tool (GPU box): where a step of decode_streams goes -- parser calls, the loop's wait for them, its GPU section and
the wait for the previous step's PCM inside it -- for a few parser thread counts.  python tools/trace_e2e.py [copies]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from libxaac_amd import decoder
    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    data = open(os.path.join(ROOT, "tests", "golden", "streams", "mix_aot29_32k.aac"), "rb").read()
    decoder.decode_streams([data] * 8)
    for threads in [int(t) for t in os.environ.get("XAAC_TRACE_THREADS", "0,8,16,24,32").split(",")]:
        best = None
        for _ in range(3):
            t = {}
            decoder.decode_streams([data] * copies, keep_pcm=False, timing=t, threads=threads, esbr=bool(int(os.environ.get("XAAC_TRACE_ESBR", "0"))))
            if best is None or t["steps_s"] < best["steps_s"]:
                best = t
        best = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in best.items()}
        best["threads"] = threads
        best["frames_per_s"] = round(best["frames"] / best["steps_s"])
        print(json.dumps(best))


if __name__ == "__main__":
    main()
