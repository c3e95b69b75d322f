#!/usr/bin/env python3
"""Build tests/golden/streams/*.aac: four short ADTS streams (AAC-LC stereo, HE-AACv1 stereo and mono, HE-AACv2) encoded by
the reference encoder (oracle/_ref/xaacenc) from a synthetic signal with clicks (short blocks, multi-envelope SBR
frames), a harmonic stack (sinusoidal coding) and level steps.  Data fixtures for tests/test_dropin_gpu.py."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_test_streams as m  # noqa: E402

ROOT = os.path.dirname(HERE)
REF = os.path.join(ROOT, "oracle", "_ref")


def main():
    out = os.path.join(ROOT, "tests", "golden", "streams")
    os.makedirs(out, exist_ok=True)
    sig = m.signals(seconds=1.6)
    x = 0.5 * sig["clicks"] + 0.35 * sig["harmonic"] + 0.3 * sig["noise_sweep"]
    wav = "/tmp/xaac_golden_mix.wav"
    m.write_wav(wav, x)
    for aot, br in ((2, 64000), (5, 48000), (29, 32000)):
        aac = os.path.join(out, "mix_aot%d_%dk.aac" % (aot, br // 1000))
        subprocess.run([os.path.join(REF, "xaacenc"), "-ifile:" + wav, "-ofile:" + aac, "-aot:%d" % aot, "-br:%d" % br,
                        "-adts:1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        print(aac, os.path.getsize(aac))
    # HE-AAC with the eSBR extension payload: harmonic SBR (sbr_patching_mode 0 frames) and inter-TES
    aac = os.path.join(out, "harm_aot5_48k.aac")
    subprocess.run([os.path.join(REF, "xaacenc"), "-ifile:" + wav, "-ofile:" + aac, "-aot:5", "-br:48000", "-adts:1", "-esbr:1",
                    "-harmonic_sbr:1", "-inter_tes_enc:1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    print(aac, os.path.getsize(aac))
    # a mono HE-AAC stream: the single-channel-element flavour of the SBR paths (one ixheaacd_sbr_dec call per frame)
    wav = "/tmp/xaac_golden_mono.wav"
    m.write_wav(wav, x[:, :1])
    aac = os.path.join(out, "mono_aot5_32k.aac")
    subprocess.run([os.path.join(REF, "xaacenc"), "-ifile:" + wav, "-ofile:" + aac, "-aot:5", "-br:32000", "-adts:1"],
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    print(aac, os.path.getsize(aac))
    # other sampling rates (other scale factor band tables, SBR frequency tables and core / SBR rate pairs): AAC-LC mono at
    # 16 kHz -- a rate at which the reference creates an SBR decoder by implicit signalling and, without payloads, never calls
    # it -- and HE-AAC stereo at 44.1 kHz; the same samples played at those rates (the encoder only sees numbers)
    import wave
    for name, fs, ch, args in (("lc_aot2_16k_mono", 16000, 1, ["-aot:2", "-br:32000"]), ("he_aot5_44k", 44100, 2, ["-aot:5", "-br:48000"])):
        wav = "/tmp/xaac_golden_%d.wav" % fs
        pcm = np.clip(np.round(x[:, :ch] * 32767.0), -32768, 32767).astype(np.int16)
        with wave.open(wav, "wb") as w:
            w.setnchannels(ch), w.setsampwidth(2), w.setframerate(fs)
            w.writeframes(pcm.tobytes())
        aac = os.path.join(out, name + ".aac")
        subprocess.run([os.path.join(REF, "xaacenc"), "-ifile:" + wav, "-ofile:" + aac, "-adts:1"] + args, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, check=True)
        print(aac, os.path.getsize(aac))
    # the other frame lengths (tests/golden/streams_ld: kept apart from the 1024-line streams the batched host tests walk):
    # AAC-LC with 960-line frames, AAC-LD and AAC-ELD with 512- and 480-line frames: raw access units behind an
    # AudioSpecificConfig (ADTS cannot signal the frame length) + the encoder's frame-size list, which the reference's test
    # decoder reads with -mp4:1 -imeta:
    out_ld = os.path.join(ROOT, "tests", "golden", "streams_ld")
    os.makedirs(out_ld, exist_ok=True)
    wav = "/tmp/xaac_golden_mix.wav"
    for name, args in (("lc960", ["-aot:2", "-br:64000", "-framesize:960"]), ("ld512", ["-aot:23", "-br:64000", "-framesize:512"]),
                       ("ld480", ["-aot:23", "-br:64000", "-framesize:480"]), ("eld512", ["-aot:39", "-br:64000", "-framesize:512"]),
                       ("eld480", ["-aot:39", "-br:64000", "-framesize:480"])):
        aac = os.path.join(out_ld, name + ".aac")
        subprocess.run([os.path.join(REF, "xaacenc"), "-ifile:" + wav, "-ofile:" + aac] + args, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, check=True)
        print(aac, os.path.getsize(aac))

    # USAC (xHE-AAC, -aot:42) streams for the eSBR seam behind the reference's own USAC front end (tests/golden/streams_usac;
    # raw access units + frame-size list like the LD streams): stereo 2:1 eSBR without / with harmonic SBR (u21, u21harm),
    # a switched FD / LPD core (u21sw), 4:1 eSBR (u41), a 768-line core with 8:3 eSBR (u83); mono streams whose LPD frames carry PVC (m21swpvc: switched,
    # ORIG_SBR and PVC_SBR frames; m21tdpvc: LPD only, every frame PVC)
    out_usac = os.path.join(ROOT, "tests", "golden", "streams_usac")
    os.makedirs(out_usac, exist_ok=True)
    for name, wav, args in (("u21", "/tmp/xaac_golden_mix.wav", ["-ccfl_idx:3"]),
                            ("u21harm", "/tmp/xaac_golden_mix.wav", ["-ccfl_idx:3", "-harmonic_sbr:1"]),
                            ("u21sw", "/tmp/xaac_golden_mix.wav", ["-ccfl_idx:3", "-usac:0"]),
                            ("u41", "/tmp/xaac_golden_mix.wav", ["-ccfl_idx:4"]),
                            ("u83", "/tmp/xaac_golden_mix.wav", ["-ccfl_idx:2"]),   # 768-line core, 8:3 eSBR
                            ("m21swpvc", "/tmp/xaac_golden_mono.wav", ["-ccfl_idx:3", "-usac:0", "-pvc_enc:1"]),
                            ("m21tdpvc", "/tmp/xaac_golden_mono.wav", ["-ccfl_idx:3", "-usac:2", "-pvc_enc:1"]),
                            # the other two SBR ratios with the tools of the 2:1 set: PVC frames at 8:3 and 4:1 (-harmonic_sbr:1 changes nothing at these ratios: the encoder's streams come out identical)
                            ("m83swpvc", "/tmp/xaac_golden_mono.wav", ["-ccfl_idx:2", "-usac:0", "-pvc_enc:1"]),
                            ("m41swpvc", "/tmp/xaac_golden_mono.wav", ["-ccfl_idx:4", "-usac:0", "-pvc_enc:1"]),
                            ("m41tdpvc", "/tmp/xaac_golden_mono.wav", ["-ccfl_idx:4", "-usac:2", "-pvc_enc:1"])):
        aac = os.path.join(out_usac, name + ".aac")
        subprocess.run([os.path.join(REF, "xaacenc"), "-ifile:" + wav, "-ofile:" + aac, "-aot:42", "-br:48000"] + args,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        print(aac, os.path.getsize(aac))

    # wider than the scope table (tests/golden/streams_wide; tests/test_dropin_gpu.py::test_wider_streams...): 5.1 streams -- the
    # reference's channel-element loop calls the same seams once per channel / element, so the drop-in serves them unchanged -- and
    # HE-AAC / HE-AACv2 with 960-line frames (the DAB+ profile: 30 QMF slots a frame; the 960-line IMDCT is the library's, the
    # 15-time-slot SBR calls stay the reference's)
    out_w = os.path.join(ROOT, "tests", "golden", "streams_wide")
    os.makedirs(out_w, exist_ok=True)
    n = x.shape[0]
    six = np.stack([x[:, 0], x[:, 1], 0.5 * (x[:, 0] + x[:, 1]), 0.3 * np.roll(x[:, 0], 700), 0.6 * np.roll(x[:, 1], 311),
                    0.6 * np.roll(x[:, 0], 1500) - 0.2 * x[:, 1]], 1)[: n * 5 // 8]
    wav6 = "/tmp/xaac_golden_6ch.wav"
    with wave.open(wav6, "wb") as w:
        w.setnchannels(6), w.setsampwidth(2), w.setframerate(48000)
        w.writeframes(np.clip(np.round(six * 32767.0), -32768, 32767).astype(np.int16).tobytes())
    for name, wav, args in (("mc6_aot2", wav6, ["-aot:2", "-br:192000", "-adts:1"]), ("mc6_aot5", wav6, ["-aot:5", "-br:128000", "-adts:1"]),
                            ("he960_aot5", "/tmp/xaac_golden_mix.wav", ["-aot:5", "-br:48000", "-framesize:960"]),
                            ("he960_aot29", "/tmp/xaac_golden_mix.wav", ["-aot:29", "-br:32000", "-framesize:960"])):
        aac = os.path.join(out_w, name + ".aac")
        subprocess.run([os.path.join(REF, "xaacenc"), "-ifile:" + wav, "-ofile:" + aac] + args, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, check=True)
        print(aac, os.path.getsize(aac))


if __name__ == "__main__":
    main()
