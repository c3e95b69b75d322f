#!/usr/bin/env python3
"""Build tests/golden/parser_ref.npz and tests/golden/decoder_ref.npz from the REAL reference decoder run on the committed ADTS
streams (tests/golden/streams/*.aac): per frame and channel the CRC32 of
  * the spectrum as the reference's Huffman decoder / inverse quantiser / scale factors leave it (entry of
    ixheaacd_channel_pair_process, channel.c:602) and as ixheaacd_imdct_process receives it (after M/S, intensity, PNS, TNS),
    with the window sequence / shape / max_sfb (oracle/ref_capture.c: XAAC_SPEC_DUMP);
  * the SBR header tables and frame data (and the PS frame) the reference holds at every ixheaacd_sbr_dec call, in the layouts of
    include/xaac_sbr.h (XAAC_CAPTURE_FILE), decoded with -esbr:0;
  * for the SBR streams the same decoded with the reference's default flags (-esbr:1, "Path A": XAAC_ESBR_SIDE_FILE): header,
    frame, the xaac_esbr_side members (include/xaac_esbr.h) and the PS frame, plus the QMF transposer's parameters
    (xaac_hbe_state) after each call's resets;
and per stream the CRC32, length and rate of the PCM `xaacdec -esbr:0` and plain `xaacdec` write.  Data only; runs only where /root/reference is
(oracle/_ref built).  The first frame's records of the initialisation pass (the reference decodes frame 0 twice) are dropped."""
import os
import subprocess
import sys
import tempfile
import wave
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
NAMES = ["mix_aot2_64k", "mix_aot5_48k", "mono_aot5_32k", "harm_aot5_48k", "mix_aot29_32k", "synth_lc_a", "synth_lc_b", "synth_lc_mono", "lc_aot2_16k_mono", "he_aot5_44k"]
REF = os.path.join(ROOT, "oracle", "_ref")


def crc(b):
    return zlib.crc32(bytes(b)) & 0xffffffff


def capture(name, tmp):
    """-> (spec records [k, 1030] int32, sbr records or [])"""
    import sbr_capture as sc
    src = os.path.join(ROOT, "tests", "golden", "streams", name + ".aac")
    spec, cap = os.path.join(tmp, "spec.bin"), os.path.join(tmp, "cap.bin")
    for f in (spec, cap):
        if os.path.exists(f):
            os.remove(f)
    env = dict(os.environ, XAAC_SPEC_DUMP=spec, XAAC_CAPTURE_FILE=cap)
    subprocess.run([os.path.join(REF, "xaacdec_capture"), "-ifile:" + src, "-ofile:" + os.path.join(tmp, "o.wav"), "-esbr:0"],
                   env=env, check=True, capture_output=True)
    raw = np.fromfile(spec, dtype=np.int32).reshape(-1, 1030)
    recs = sc.read_records(cap) if os.path.exists(cap) and os.path.getsize(cap) else []
    return raw, recs


def capture_esbr(name, tmp):
    """-> list of (meta int32[8], header, frame, esbr_side, ps_frame bytes, hbe parameters int32[11]) per ixheaacd_sbr_dec call"""
    import ctypes
    import sbr_capture as sc
    from esbr_structs import EsbrSide
    src = os.path.join(ROOT, "tests", "golden", "streams", name + ".aac")
    cap = os.path.join(tmp, "esbr.bin")
    if os.path.exists(cap):
        os.remove(cap)
    subprocess.run([os.path.join(REF, "xaacdec_capture"), "-ifile:" + src, "-ofile:" + os.path.join(tmp, "o.wav")],
                   env=dict(os.environ, XAAC_ESBR_SIDE_FILE=cap), check=True, capture_output=True)
    raw = open(cap, "rb").read()
    sizes = [32, ctypes.sizeof(sc.Header), ctypes.sizeof(sc.Frame), ctypes.sizeof(EsbrSide), ctypes.sizeof(sc.PsFrame), 44]
    rs = sum(sizes)
    assert len(raw) % rs == 0
    out = []
    for o in range(0, len(raw), rs):
        parts, p = [], o
        for z in sizes:
            parts.append(raw[p:p + z])
            p += z
        meta = np.frombuffer(parts[0], np.int32)
        assert meta[0] == 0x58414531 and meta[4] == 1
        out.append((meta, parts[1], parts[2], parts[3], parts[4], np.frombuffer(parts[5], np.int32)))
    return out


def main():
    d, dec = {}, {"crc": [], "samples": [], "rate": [], "crc_esbr": [], "samples_esbr": []}
    with tempfile.TemporaryDirectory() as tmp:
        for name in NAMES:
            raw, recs = capture(name, tmp)
            t1, t2 = raw[raw[:, 0] == 1], raw[raw[:, 0] == 2]
            n_ch = 2 if t1[1, 1] == 1 else 1
            t1, t2 = t1[n_ch:], t2[n_ch:]                       # drop the initialisation pass over frame 0
            frames = len(t1) // n_ch
            core = np.zeros((frames, n_ch, 5), np.uint32)       # crc before tools, crc at the IMDCT, sequence, shape, max_sfb
            for k in range(frames * n_ch):
                f, c = divmod(k, n_ch)
                assert tuple(t1[k, 2:5]) == tuple(t2[k, 2:5])
                core[f, c] = crc(t1[k, 6:].tobytes()), crc(t2[k, 6:].tobytes()), t1[k, 2], t1[k, 3], t1[k, 4]
            d[name + "_core"] = core
            if recs:
                recs = recs[n_ch:]
                assert len(recs) == frames * n_ch, (name, len(recs), frames)
                side = np.zeros((frames, n_ch, 3), np.uint32)   # crc of header, of frame, of the PS frame (0 without PS)
                for k, r in enumerate(recs):
                    f, c = divmod(k, n_ch)
                    side[f, c] = crc(r["header"]), crc(r["frame"]), crc(r["ps_frame"]) if r["ps"] else 0
                d[name + "_sbr"] = side
                er = capture_esbr(name, tmp)[n_ch:]
                assert len(er) == frames * n_ch, (name, len(er), frames)
                eside = np.zeros((frames, n_ch, 4), np.uint32)  # crc of header, frame, esbr side, PS frame (0 without PS)
                hbe = np.zeros((frames, n_ch, 11), np.int32)    # synth_size k_start start_band end_band x_over_qmf[6] max_stretch
                for k, (meta, h, f_, e, p_, par) in enumerate(er):
                    f, c = divmod(k, n_ch)
                    eside[f, c] = crc(h), crc(f_), crc(e), crc(p_) if meta[2] else 0
                    hbe[f, c] = par
                d[name + "_esbr"], d[name + "_hbe"] = eside, hbe
            out = os.path.join(tmp, name + ".wav")
            subprocess.run([os.path.join(REF, "xaacdec"), "-ifile:" + os.path.join(ROOT, "tests", "golden", "streams", name + ".aac"),
                            "-ofile:" + out, "-esbr:0"], check=True, capture_output=True)
            with wave.open(out) as w:
                pcm = w.readframes(w.getnframes())
                dec["crc"].append(crc(pcm)), dec["samples"].append(w.getnframes()), dec["rate"].append(w.getframerate())
            subprocess.run([os.path.join(REF, "xaacdec"), "-ifile:" + os.path.join(ROOT, "tests", "golden", "streams", name + ".aac"),
                            "-ofile:" + out], check=True, capture_output=True)
            with wave.open(out) as w:
                dec["crc_esbr"].append(crc(w.readframes(w.getnframes()))), dec["samples_esbr"].append(w.getnframes())
            print(name, "frames", frames, "channels", n_ch, "sbr records", len(recs), "pcm samples", dec["samples"][-1])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "parser_ref.npz"), **d)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "decoder_ref.npz"), crc=np.array(dec["crc"], np.uint32),
                        samples=np.array(dec["samples"], np.int64), rate=np.array(dec["rate"], np.int64),
                        crc_esbr=np.array(dec["crc_esbr"], np.uint32), samples_esbr=np.array(dec["samples_esbr"], np.int64))


if __name__ == "__main__":
    main()
