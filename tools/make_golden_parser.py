#!/usr/bin/env python3
"""Build tests/golden/parser_ref.npz and tests/golden/decoder_ref.npz from the REAL reference decoder run on the committed ADTS
streams (tests/golden/streams/*.aac): per frame and channel the CRC32 of
  * the spectrum as the reference's Huffman decoder / inverse quantiser / scale factors leave it (entry of
    ixheaacd_channel_pair_process, channel.c:602) and as ixheaacd_imdct_process receives it (after M/S, intensity, PNS, TNS),
    with the window sequence / shape / max_sfb (oracle/ref_capture.c: XAAC_SPEC_DUMP);
  * the SBR header tables and frame data (and the PS frame) the reference holds at every ixheaacd_sbr_dec call, in the layouts of
    include/xaac_sbr.h (XAAC_CAPTURE_FILE), decoded with -esbr:0;
and per stream the CRC32, length and rate of the PCM `xaacdec -esbr:0` writes.  Data only; runs only where /root/reference is
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
NAMES = ["mix_aot2_64k", "mix_aot5_48k", "mono_aot5_32k", "harm_aot5_48k", "mix_aot29_32k", "synth_lc_a", "synth_lc_b", "synth_lc_mono"]
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


def main():
    d, dec = {}, {"crc": [], "samples": [], "rate": []}
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
            out = os.path.join(tmp, name + ".wav")
            subprocess.run([os.path.join(REF, "xaacdec"), "-ifile:" + os.path.join(ROOT, "tests", "golden", "streams", name + ".aac"),
                            "-ofile:" + out, "-esbr:0"], check=True, capture_output=True)
            with wave.open(out) as w:
                pcm = w.readframes(w.getnframes())
                dec["crc"].append(crc(pcm)), dec["samples"].append(w.getnframes()), dec["rate"].append(w.getframerate())
            print(name, "frames", frames, "channels", n_ch, "sbr records", len(recs), "pcm samples", dec["samples"][-1])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "parser_ref.npz"), **d)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "decoder_ref.npz"), crc=np.array(dec["crc"], np.uint32),
                        samples=np.array(dec["samples"], np.int64), rate=np.array(dec["rate"], np.int64))


if __name__ == "__main__":
    main()
