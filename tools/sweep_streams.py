#!/usr/bin/env python3
"""Parity sweep beyond the committed streams (GPU box; oracle/_ref must have travelled): the reference encoder makes ADTS
streams at several sampling rates, bit rates, channel counts and object types from the synthetic test signal; every stream
is decoded by the reference (`xaacdec`, default flags and -esbr:0) and by the repo's native decoder
(libxaac_amd/xaacdec_amd, same flags); the WAV payloads must be identical.  Prints one line per (stream, flag)."""
import os
import subprocess
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_test_streams as m  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
CLI = os.path.join(ROOT, "libxaac_amd", "xaacdec_amd")
TMP = os.environ.get("SWEEP_TMP", "/tmp/xaac_sweep")


def payload(path):
    with wave.open(path) as w:
        return w.getnchannels(), w.getframerate(), w.readframes(w.getnframes())


def main():
    os.makedirs(TMP, exist_ok=True)
    sig = m.signals(seconds=float(os.environ.get("SWEEP_SECONDS", "1.6")))   # (multiples of 0.8 s)
    x48 = 0.5 * sig["clicks"] + 0.35 * sig["harmonic"] + 0.3 * sig["noise_sweep"]
    variant = int(os.environ.get("SWEEP_VARIANT", "0"))   # other mixes of the same ingredients: weights, time offsets, polarity
    if variant:
        vr = np.random.default_rng(1000 + variant)
        w = vr.uniform(0.05, 0.9, 3)
        x48 = sum(wk * np.roll(sig[k], int(vr.integers(0, 48000)), axis=0) * vr.choice([-1.0, 1.0]) for wk, k in zip(w, ("clicks", "harmonic", "noise_sweep")))
        x48 = x48 + vr.uniform(0.0, 0.2) * vr.standard_normal(x48.shape)
        x48[:, 1] = vr.uniform(0.2, 1.0) * x48[:, 1] + vr.uniform(0.0, 0.5) * np.roll(x48[:, 0], int(vr.integers(1, 400)))
        x48 = np.clip(x48 / max(1.0, np.abs(x48).max() / 0.98), -1.0, 1.0)
    cases = []
    rates = tuple(int(v) for v in os.environ["SWEEP_RATES"].split(",")) if os.environ.get("SWEEP_RATES") else (16000, 22050, 24000, 32000, 44100, 48000)
    for fs in rates:
        # the same samples played at another rate: the encoder only sees numbers
        for ch in (1, 2):
            wav = os.path.join(TMP, "in_%d_%d.wav" % (fs, ch))
            pcm = np.clip(np.round(x48[:, :ch] * 32767.0), -32768, 32767).astype(np.int16)
            with wave.open(wav, "wb") as w:
                w.setnchannels(ch); w.setsampwidth(2); w.setframerate(fs)
                w.writeframes(pcm.tobytes())
            for aot, brs in ((2, (32000, 96000)), (5, (24000, 48000)), (29, (18000, 32000))):
                if aot == 29 and ch != 2:
                    continue       # parametric stereo codes a stereo input
                if aot != 2 and fs < 32000 and not os.environ.get("SWEEP_ALL_SBR_RATES"):
                    continue       # SBR at twice a low core rate: the encoder's supported range
                for br in brs:
                    cases.append((fs, ch, aot, br, wav))
                    if aot == 5:   # ... and with ENHSBR elements: harmonic patching, pre-flattening, inter-TES
                        cases.append((fs, ch, aot, br, wav, ("-esbr:1", "-harmonic_sbr:1", "-inter_tes_enc:1")))
    # other material at one rate: digital silence, full-scale noise (the peak limiter at work, escape codes, saturating float
    # samples on Path A), a lone click train; stereo
    rng = np.random.default_rng(3)
    n = x48.shape[0]
    extra = {"silence": np.zeros((n, 2)), "loud": np.clip(rng.standard_normal((n, 2)) * 0.9, -1.0, 1.0),
             "clicks": np.clip(1.9 * sig["clicks"], -1.0, 1.0)}
    for label, x in extra.items():
        wav = os.path.join(TMP, "in_%s.wav" % label)
        pcm = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
        with wave.open(wav, "wb") as w:
            w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100)
            w.writeframes(pcm.tobytes())
        for aot, br in ((2, 64000), (5, 32000), (29, 24000)):
            cases.append((label, 2, aot, br, wav))
    bad = total = 0
    for fs, ch, aot, br, wav, *enc_extra in cases:
        enc_extra = enc_extra[0] if enc_extra else ()
        name = "s%s_c%d_a%d_b%d%s" % (fs, ch, aot, br, "_enh" if enc_extra else "")
        aac = os.path.join(TMP, name + ".aac")
        # SWEEP_ENC_EXTRA: encoder options for every stream of the run, e.g. "-tns:0", "-full_bandwidth:1", "-max_out_buffer_per_ch:6144"
        r = subprocess.run([os.path.join(REF, "xaacenc"), "-ifile:" + wav, "-ofile:" + aac, "-aot:%d" % aot, "-br:%d" % br, "-adts:1", *enc_extra,
                            *os.environ.get("SWEEP_ENC_EXTRA", "").split()],
                           capture_output=True)
        if r.returncode or not os.path.exists(aac) or os.path.getsize(aac) < 100:
            print(name, "encoder refused")
            continue
        dropin = os.environ.get("SWEEP_DECODER") == "dropin"   # the reference decoder with its seams served by the library
        flag_sets = ((), ("-esbr:0",), ("-dsample:1",), ("-dsample:1", "-esbr:0")) if dropin else ((), ("-esbr:0",))
        tol = int(os.environ.get("SWEEP_TOL", "0"))   # largest sample difference taken as equal (float paths: +-1 LSB)
        if os.environ.get("SWEEP_FLAGS"):             # one flag set instead, e.g. "-esbr_hq:1"
            flag_sets = (tuple(os.environ["SWEEP_FLAGS"].split()),)
            if aot == 2:
                continue
        for flags in flag_sets:
            a, b = os.path.join(TMP, "ref.wav"), os.path.join(TMP, "own.wav")
            for f in (a, b):
                if os.path.exists(f):
                    os.remove(f)
            r1 = subprocess.run([os.path.join(REF, "xaacdec"), "-ifile:" + aac, "-ofile:" + a, *flags], capture_output=True)
            if dropin:
                r2 = subprocess.run([os.path.join(REF, "xaacdec_dropin"), "-ifile:" + aac, "-ofile:" + b, *flags], capture_output=True, text=True)
            else:
                r2 = subprocess.run([CLI, "-ifile:" + aac, "-ofile:" + b, "-quiet", *flags], capture_output=True, text=True)
            total += 1
            if r1.returncode or not os.path.exists(a) or len(payload(a)[2]) == 0:
                # (e.g. -esbr_hq:1 on a stream whose transposer sizes the reference has no transforms for: it writes no sample)
                print(name, flags, "reference decoder failed or wrote nothing: not compared")
                continue
            if r2.returncode or not os.path.exists(b):
                print(name, flags, "OWN DECODER FAILED:", r2.stderr.strip()[-200:])
                bad += 1
                continue
            pa, pb = payload(a), payload(b)
            if pa == pb:
                print(name, flags, "identical", pa[0], "ch", pa[1], "Hz", len(pa[2]) // (2 * pa[0]), "samples")
            elif tol and pa[:2] == pb[:2] and len(pa[2]) == len(pb[2]) and \
                    int(np.abs(np.frombuffer(pa[2], np.int16).astype(int) - np.frombuffer(pb[2], np.int16)).max()) <= tol:
                nd = int((np.frombuffer(pa[2], np.int16) != np.frombuffer(pb[2], np.int16)).sum())
                print(name, flags, "identical within", tol, "LSB:", nd, "of", len(pa[2]) // 2, "samples differ")
            else:
                bad += 1
                if pa[:2] != pb[:2] or len(pa[2]) != len(pb[2]):
                    print(name, flags, "DIFFERENT SHAPE", pa[:2], len(pa[2]), pb[:2], len(pb[2]))
                else:
                    x = np.frombuffer(pa[2], np.int16).reshape(-1, pa[0]); y = np.frombuffer(pb[2], np.int16).reshape(-1, pa[0])
                    d = np.nonzero(np.any(x != y, axis=1))[0]
                    print(name, flags, "DIFFERENT", d.size, "samples, first", int(d[0]), "max", int(np.abs(x.astype(int) - y).max()))
    print("cases", total, "bad", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
