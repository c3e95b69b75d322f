#!/usr/bin/env python3
"""Independent streams whose SBR headers change at DIFFERENT frames, decoded in one batch (GPU box): parts of the same rate
and channel count but different bit rates (different SBR ranges: a new SBR header, i.e. a reset of the SBR decoder, where
one part follows another) are spliced frame-wise at different places per file; the native decoder takes the files as one
-ilist batch -- most steps see no reset, some see one or two streams reset while the others go on -- with -esbr:0 and with
the reference's default flags (Path A: the reset-time transposer runs on those streams' rows alone).  Every WAV must equal
what the reference decoder writes for that file alone."""
import os, subprocess, sys, wave
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_test_streams as m
REF = os.path.join(ROOT, "oracle", "_ref"); CLI = os.path.join(ROOT, "libxaac_amd", "xaacdec_amd")
TMP = os.environ.get("SWEEP_TMP", "/tmp/xaac_splice_list"); os.makedirs(TMP, exist_ok=True)
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
def frames(data):
    pos, out = 0, []
    while pos + 7 <= len(data):
        n = ((data[pos + 3] & 3) << 11) | (data[pos + 4] << 3) | (data[pos + 5] >> 5)
        out.append(data[pos:pos + n]); pos += n
    return out
def payload(path):
    with wave.open(path) as w:
        return w.getnchannels(), w.getframerate(), w.readframes(w.getnframes())
HARM = ("-esbr:1", "-harmonic_sbr:1", "-inter_tes_enc:1")
CASES = [("2_5", [(2, 5, br, ()) for br in (24000, 64000, 32000)]), ("1_5", [(1, 5, br, ()) for br in (16000, 40000, 24000)]),
         ("2_29", [(2, 29, br, ()) for br in (18000, 40000, 24000)]), ("2_5_harmonic", [(2, 5, br, HARM) for br in (48000, 32000, 64000)])]
bad = 0
for label, plist in CASES:
    p1, p2, p3 = (frames(enc("q%d_%d_%d_%d" % (c, a, br, len(extra)), c, a, br, extra)) for c, a, br, extra in plist)
    files = {"a": p1 + p2 + p3, "b": p1[:10] + p2 + p3[:20], "c": list(p2), "d": p1[:25] + p3, "e": p3[:4] + p1[:9] + p2[:30]}
    d = os.path.join(TMP, label); os.makedirs(d, exist_ok=True)
    for n, fr in files.items():
        open(os.path.join(d, n + ".aac"), "wb").write(b"".join(fr))
    lst = os.path.join(d, "list.txt")
    open(lst, "w").write("\n".join(os.path.join(d, n + ".aac") for n in files) + "\n")
    for flags in (("-esbr:0",), ()):
        out = os.path.join(d, "out" + ("0" if flags else "1")); os.makedirs(out, exist_ok=True)
        r = subprocess.run([CLI, "-ilist:" + lst, "-odir:" + out, "-quiet", *flags], capture_output=True, text=True)
        if r.returncode:
            print(label, flags, "own decoder:", r.stderr.strip()[-160:]); bad += 1; continue
        for n in files:
            ref = os.path.join(d, n + "_ref.wav")
            subprocess.run([os.path.join(REF, "xaacdec"), "-ifile:" + os.path.join(d, n + ".aac"), "-ofile:" + ref, *flags], capture_output=True)
            pa, pb = payload(ref), payload(os.path.join(out, n + ".wav"))
            if pa == pb: print(label, n, flags, "identical", len(pa[2]) // (2 * pa[0]), "samples")
            else:
                bad += 1
                if len(pa[2]) != len(pb[2]): print(label, n, flags, "DIFFERENT LENGTH", len(pa[2]), len(pb[2]))
                else:
                    u = np.frombuffer(pa[2], np.int16).reshape(-1, pa[0]); v = np.frombuffer(pb[2], np.int16).reshape(-1, pa[0])
                    dd = np.nonzero(np.any(u != v, axis=1))[0]
                    print(label, n, flags, "DIFFERENT", dd.size, "samples, first", int(dd[0]), "of", len(u))
print("bad", bad)
