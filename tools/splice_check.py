#!/usr/bin/env python3
"""Mid-stream configuration changes (GPU box): ADTS streams of the same rate and channel count but different bit rates --
different SBR frequency ranges, i.e. a new SBR header in the middle -- are spliced frame-wise and decoded by the reference
and by the native decoder (and by libxaac_amd/decoder.py) with -esbr:0 and with the default flags; parts encoded with ENHSBR
elements (harmonic patching, pre-flattening) make the transposer's state behind a mid-stream reset audible."""
import os, subprocess, sys, wave
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_test_streams as m
REF = os.path.join(ROOT, "oracle", "_ref"); CLI = os.path.join(ROOT, "libxaac_amd", "xaacdec_amd")
TMP = os.environ.get("SWEEP_TMP", "/tmp/xaac_splice"); os.makedirs(TMP, exist_ok=True)
sig = m.signals(seconds=1.6)
x = 0.5 * sig["clicks"] + 0.35 * sig["harmonic"] + 0.3 * sig["noise_sweep"]
def enc(name, ch, aot, br, extra=()):
    wav = os.path.join(TMP, "in%d.wav" % ch)
    pcm = np.clip(np.round(x[:, :ch] * 32767.0), -32768, 32767).astype(np.int16)
    with wave.open(wav, "wb") as w:
        w.setnchannels(ch); w.setsampwidth(2); w.setframerate(48000); w.writeframes(pcm.tobytes())
    aac = os.path.join(TMP, name + ".aac")
    subprocess.run([os.path.join(REF, "xaacenc"), "-ifile:" + wav, "-ofile:" + aac, "-aot:%d" % aot, "-br:%d" % br, "-adts:1", *extra], capture_output=True, check=True)
    return open(aac, "rb").read()
def payload(path):
    with wave.open(path) as w:
        return w.getnchannels(), w.getframerate(), w.readframes(w.getnframes())
bad = 0
HARM = ("-esbr:1", "-harmonic_sbr:1", "-inter_tes_enc:1")   # ENHSBR elements: harmonic patching (the transposer's output is read), pre-flattening
CASES = [("2 5", [(2, 5, br, ()) for br in (24000, 64000, 32000)]), ("1 5", [(1, 5, br, ()) for br in (16000, 40000)]),
         ("2 29", [(2, 29, br, ()) for br in (18000, 40000, 24000)]), ("2 2", [(2, 2, br, ()) for br in (48000, 128000)]),
         ("2 5 harmonic", [(2, 5, br, HARM) for br in (48000, 32000, 64000)]), ("1 5 harmonic", [(1, 5, br, HARM) for br in (32000, 20000)]),
         # parametric stereo appearing and disappearing on a mono core
         ("mono + ps + mono", [(1, 5, 24000, ()), (2, 29, 24000, ()), (1, 5, 32000, ())]), ("ps + mono", [(2, 29, 32000, ()), (1, 5, 24000, ())])]
for label, plist in CASES:
    ch, aot = label, ""
    parts = [enc("p%d_%d_%d_%d" % (c, a, br, len(extra)), c, a, br, extra) for c, a, br, extra in plist]
    spliced = os.path.join(TMP, "splice_%s.aac" % label.replace(" ", "_").replace("+", "p"))
    open(spliced, "wb").write(b"".join(parts))
    for flags in (("-esbr:0",), ()):
        a, b = os.path.join(TMP, "ref.wav"), os.path.join(TMP, "own.wav")
        for f in (a, b):
            if os.path.exists(f): os.remove(f)
        r1 = subprocess.run([os.path.join(REF, "xaacdec"), "-ifile:" + spliced, "-ofile:" + a, *flags], capture_output=True)
        r2 = subprocess.run([CLI, "-ifile:" + spliced, "-ofile:" + b, "-quiet", *flags], capture_output=True, text=True)
        if r2.returncode:
            print(ch, aot, flags, "own decoder:", r2.stderr.strip()[-120:]); bad += 1; continue
        pa, pb = payload(a), payload(b)
        if os.environ.get("SPLICE_PYTHON", "1") != "0":   # ... and libxaac_amd/decoder.py: the same loop in Python
            sys.path.insert(0, ROOT)
            from libxaac_amd import decoder
            got, rate = decoder.decode_streams([open(spliced, "rb").read()] * 2, esbr=not flags)
            want = np.frombuffer(pa[2], np.int16).reshape(-1, pa[0])
            if rate != pa[1] or any(g.shape != want.shape or not np.array_equal(g, want) for g in got):
                print(ch, aot, flags, "decode_streams DIFFERENT")
                bad += 1
        if pa == pb: print(ch, aot, flags, "identical", len(pa[2]) // (2 * pa[0]), "samples")
        else:
            bad += 1
            if len(pa[2]) != len(pb[2]): print(ch, aot, flags, "DIFFERENT LENGTH", len(pa[2]), len(pb[2]))
            else:
                u = np.frombuffer(pa[2], np.int16).reshape(-1, pa[0]); v = np.frombuffer(pb[2], np.int16).reshape(-1, pa[0])
                d = np.nonzero(np.any(u != v, axis=1))[0]
                print(ch, aot, flags, "DIFFERENT", d.size, "samples, first", int(d[0]), "of", len(u))
print("bad", bad)
