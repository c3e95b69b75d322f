#!/usr/bin/env python3
"""USAC (xHE-AAC) streams made on the spot through the drop-in (GPU box; oracle/_ref must have travelled): the reference encoder
(-aot:42) at three sampling rates, mono and stereo, two bit rates, with its core modes (FD, switched, TD), the three SBR ratios and no
SBR, harmonic SBR, PVC, inter-TES, complex prediction and noise filling; every stream decoded by the reference decoder and by the
reference decoder with its seams served by the library (xaacdec_dropin: ixheaacd_fd_frm_dec, ixheaacd_sbr_dec); the WAVs must be
byte-identical.  Prints one line per stream."""
import os
import subprocess
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_test_streams as m  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
TMP = os.environ.get("SWEEP_TMP", "/tmp/xaac_sweep_usac")
CONFIGS = [("fd21", ["-ccfl_idx:3", "-usac:1"]), ("sw21", ["-ccfl_idx:3", "-usac:0"]), ("td21", ["-ccfl_idx:3", "-usac:2"]),
           ("fd83", ["-ccfl_idx:2", "-usac:1"]), ("sw83", ["-ccfl_idx:2", "-usac:0"]), ("fd41", ["-ccfl_idx:4", "-usac:1"]),
           ("sw41", ["-ccfl_idx:4", "-usac:0"]), ("fd21harm", ["-ccfl_idx:3", "-usac:1", "-harmonic_sbr:1"]),
           ("sw21pvc", ["-ccfl_idx:3", "-usac:0", "-pvc_enc:1"]), ("fd21tes", ["-ccfl_idx:3", "-usac:1", "-inter_tes_enc:1"]),
           ("fd1024", ["-ccfl_idx:1", "-usac:1"]), ("sw768", ["-ccfl_idx:0", "-usac:0"]),
           ("fd21cplx", ["-ccfl_idx:3", "-usac:1", "-cmpx_pred:1"]), ("fd21nf", ["-ccfl_idx:3", "-usac:1", "-nf:1"])]
CONFIGS = [(n, ["-aot:42"] + a) for n, a in CONFIGS]
# SWEEP_PROFILE=ld: the other frame lengths instead -- AAC-LD / AAC-ELD with 512- and 480-line frames (ELD brings its low-delay SBR
# where the encoder turns it on), AAC-LC / HE-AAC / HE-AACv2 with 960-line frames (raw access units behind an AudioSpecificConfig)
LD_CONFIGS = [("ld512", ["-aot:23", "-framesize:512"]), ("ld480", ["-aot:23", "-framesize:480"]), ("eld512", ["-aot:39", "-framesize:512"]),
              ("eld480", ["-aot:39", "-framesize:480"]), ("lc960", ["-aot:2", "-framesize:960"]), ("he960", ["-aot:5", "-framesize:960"]),
              ("hev2_960", ["-aot:29", "-framesize:960"])]


def main():
    os.makedirs(TMP, exist_ok=True)
    sig = m.signals(seconds=float(os.environ.get("SWEEP_SECONDS", "1.6")))
    x48 = 0.5 * sig["clicks"] + 0.35 * sig["harmonic"] + 0.3 * sig["noise_sweep"]
    total = bad = refused = 0
    gpu_calls = 0
    rates = tuple(int(v) for v in os.environ.get("SWEEP_RATES", "32000,44100,48000").split(","))
    for fs in rates:
        for ch in (1, 2):
            wav = os.path.join(TMP, "in_%d_%d.wav" % (fs, ch))
            pcm = np.clip(np.round(x48[:, :ch] * 32767.0), -32768, 32767).astype(np.int16)
            with wave.open(wav, "wb") as w:
                w.setnchannels(ch); w.setsampwidth(2); w.setframerate(fs)
                w.writeframes(pcm.tobytes())
            for br in ((24000, 64000) if ch == 1 else (32000, 96000)):
                for name, args in (LD_CONFIGS if os.environ.get("SWEEP_PROFILE") == "ld" else CONFIGS):
                    if name == "hev2_960" and ch != 2:
                        continue
                    tag = "u%d_c%d_b%d_%s" % (fs, ch, br, name)
                    aac = os.path.join(TMP, tag + ".aac")
                    for f in (aac, aac[:-4] + ".txt"):
                        if os.path.exists(f):
                            os.remove(f)
                    r = subprocess.run([os.path.join(REF, "xaacenc"), "-ifile:" + wav, "-ofile:" + aac, "-br:%d" % br] + args,
                                       capture_output=True)
                    if r.returncode or not os.path.exists(aac) or os.path.getsize(aac) < 100 or not os.path.exists(aac[:-4] + ".txt"):
                        print(tag, "encoder refused")
                        refused += 1
                        continue
                    ex = ["-mp4:1", "-imeta:" + aac[:-4] + ".txt"]
                    a, b = os.path.join(TMP, "ref.wav"), os.path.join(TMP, "gpu.wav")
                    for f in (a, b):
                        if os.path.exists(f):
                            os.remove(f)
                    subprocess.run([os.path.join(REF, "xaacdec"), "-ifile:" + aac, "-ofile:" + a] + ex, capture_output=True)
                    r2 = subprocess.run([os.path.join(REF, "xaacdec_dropin"), "-ifile:" + aac, "-ofile:" + b] + ex, capture_output=True, text=True)
                    if not os.path.exists(a) or os.path.getsize(a) < 1000:
                        print(tag, "reference decoder wrote nothing")
                        continue
                    total += 1
                    import re
                    mm = re.search(r"(\d+) USAC fd_frm_dec calls ran on the GPU", r2.stderr)
                    ms = re.search(r"(\d+) of them for USAC channels, (\d+) sbr_dec calls left", r2.stderr)
                    calls = (int(mm.group(1)) if mm else 0, int(ms.group(1)) if ms else 0, int(ms.group(2)) if ms else -1)
                    if os.environ.get("SWEEP_PROFILE") == "ld":   # (IMDCT calls of 960-line + LD / ELD frames, whole low-delay SBR calls)
                        m1 = re.search(r"(\d+) imdct_process calls of 960-line frames and (\d+) of AAC-LD / ELD frames", r2.stderr)
                        m2 = re.search(r"(\d+) whole low-delay SBR calls", r2.stderr)
                        calls = (int(m1.group(1)) + int(m1.group(2)) if m1 else 0, int(m2.group(1)) if m2 else 0, calls[2])
                    gpu_calls += calls[0] + calls[1]
                    same = os.path.exists(b) and open(a, "rb").read() == open(b, "rb").read()
                    bad += not same
                    print(tag, "identical" if same else "DIFFERENT", "fd / sbr calls on the GPU %d / %d, sbr left to the reference %d" % calls)
    print("cases", total, "bad", bad, "refused", refused, "gpu_calls", gpu_calls)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
