#!/usr/bin/env python3
"""Synthesise test signals, encode them with the reference encoder (oracle/_ref/xaacenc) and decode them
with the capture build of the reference decoder (oracle/_ref/xaacdec_capture, -esbr:0) so that the captured
ixheaacd_sbr_dec calls cover multi-envelope / transient frames, sinusoidal coding, several frequency tables.
Only runs where oracle/_ref exists.  Output: <outdir>/<name>.{wav,aac,cap}."""
import os
import subprocess
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
FS = 48000


def write_wav(path, x):
    x = np.clip(x, -1, 1)
    pcm = (x * 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(pcm.shape[1]); w.setsampwidth(2); w.setframerate(FS)
        w.writeframes(pcm.tobytes())


def signals(seconds=4.0):
    n = int(FS * seconds)
    t = np.arange(n) / FS
    rng = np.random.default_rng(1234)
    out = {}
    # castanet-like clicks over a pad: transients -> VARFIX/FIXVAR frames, several envelopes
    x = 0.1 * np.sin(2 * np.pi * 220 * t)
    for p in np.arange(0.13, seconds, 0.171):
        i = int(p * FS)
        env = np.exp(-np.arange(2400) / 150.0)
        x[i:i + 2400] += 0.8 * env * rng.standard_normal(2400)[: len(x[i:i + 2400])]
    out["clicks"] = np.stack([x, np.roll(x, 700)], 1)
    # harmonic-rich tones reaching into the SBR range -> add_harmonics
    x = np.zeros(n)
    for k in range(1, 40):
        x += (0.5 / k) * np.sin(2 * np.pi * 523.25 * k * t + k)
    x *= 0.5 * (1 + np.sin(2 * np.pi * 0.7 * t))
    y = sum(0.2 * np.sin(2 * np.pi * f * t) for f in (9000.0, 11000.0, 13500.0, 16000.0))
    out["harmonic"] = np.stack([0.4 * x + y, 0.4 * x], 1)
    # noise with level steps and a sweep
    nz = rng.standard_normal(n) * (0.05 + 0.3 * (np.floor(t * 3) % 2))
    sw = 0.3 * np.sin(2 * np.pi * (200 * t + 2400 * t * t))
    out["noise_sweep"] = np.stack([nz + sw, nz[::-1] * 0.5 + sw], 1)
    return out


def main(outdir, only_aot=None):
    os.makedirs(outdir, exist_ok=True)
    made = []
    for name, x in signals().items():
        wav = os.path.join(outdir, name + ".wav")
        write_wav(wav, x)
        for aot, br in ((5, 32000), (5, 48000), (5, 64000), (29, 24000), (29, 32000)):
            if only_aot is not None and aot != only_aot:
                continue
            tag = "%s_aot%d_%dk" % (name, aot, br // 1000)
            aac = os.path.join(outdir, tag + ".aac")
            cap = os.path.join(outdir, tag + ".cap")
            subprocess.run([os.path.join(REF, "xaacenc"), "-ifile:" + wav, "-ofile:" + aac, "-aot:%d" % aot,
                            "-br:%d" % br, "-adts:1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            env = dict(os.environ, XAAC_CAPTURE_FILE=cap)
            subprocess.run([os.path.join(REF, "xaacdec_capture"), "-ifile:" + aac, "-ofile:" + aac + ".wav", "-esbr:0"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
            if os.path.exists(cap):
                made.append(cap)
    print("\n".join(made))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/xaac_streams", int(sys.argv[2]) if len(sys.argv) > 2 else None)
