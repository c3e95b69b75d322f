#!/usr/bin/env python3
"""debugging aid: own decoder in -esbr:1 mode against oracle/_ref/xaacdec (default flags), frame by frame"""
import os, subprocess, sys, wave
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libxaac_amd import decoder
for name in sys.argv[1:] or ["mix_aot5_48k", "mono_aot5_32k", "harm_aot5_48k", "mix_aot29_32k"]:
    src = os.path.join(ROOT, "tests/golden/streams", name + ".aac")
    out = "/tmp/%s.wav" % name
    subprocess.run([os.path.join(ROOT, "oracle/_ref/xaacdec"), "-ifile:" + src, "-ofile:" + out], check=True, capture_output=True)
    with wave.open(out) as w:
        want = np.frombuffer(w.readframes(w.getnframes()), np.int16).reshape(-1, w.getnchannels())
    got, rate = decoder.decode_streams([open(src, "rb").read()], esbr=True)
    g = got[0]
    print(name, "want", want.shape, "got", g.shape, rate)
    m = min(len(g), len(want))
    d = np.abs(g[:m].astype(np.int32) - want[:m].astype(np.int32))
    for f in range(m // 2048):
        b = d[f * 2048:(f + 1) * 2048]
        if b.any():
            print("  frame", f, "differing per channel", (b > 0).sum(0).tolist(), "max", b.max(0).tolist(), "first", int(np.nonzero(b.any(1))[0][0]))
    print("  total differing", int((d > 0).sum()))
