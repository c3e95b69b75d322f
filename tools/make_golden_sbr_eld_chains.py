#!/usr/bin/env python3
"""Build tests/golden/sbr_eld_chains.npz: steps of the REAL ixheaacd_sbr_dec run as an AAC-ELD channel's (low-delay SBR:
audio_object_type AOT_ER_AAC_ELD, HQ, 16 or 15 QMF slots a frame, the LD complex banks), as chains with the state carried by
the reference itself (oracle/ref_sbr_adapter.c: ref_sbr_dec_eld drives the compiled reference).

The side info is what the reference decoder held while it decoded the committed AAC-ELD streams (tests/golden/streams_ld/eld512,
eld480: oracle/_ref/xaacdec_capture), walked in order from the stream's start with reference-side fuzz on most steps (inverse-
filter modes, limiter gains, interpolation, smoothing, added harmonics, envelope exponents).  The core PCM of a step is the
counter-based generator of tools/make_golden_sbr_chains.py (kind 3), 32 samples per slot.  Stored per step: header, frame, the
return code, CRC32s of the PCM, of the state after the call and of the rows the synthesis bank hands on; per chain the state at
its start.  Data only; runs only where /root/reference exists."""
import ctypes
import os
import subprocess
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import sbr_capture as c  # noqa: E402
from make_golden_sbr_chains import chain_pcm, fuzz_sbr  # noqa: E402

P16, P32 = ctypes.POINTER(ctypes.c_int16), ctypes.POINTER(ctypes.c_int32)
STREAMS = ("eld512", "eld480")
PASSES = 4


def crc(b):
    return zlib.crc32(bytes(b)) & 0xffffffff


def captured(name):
    src = os.path.join(ROOT, "tests", "golden", "streams_ld", name + ".aac")
    tmp = "/tmp/xaac_eld_cap_%s.bin" % name
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "xaacdec_capture"), "-ifile:" + src, "-ofile:/tmp/xaac_eld_cap.wav", "-mp4:1",
                    "-imeta:" + src[:-4] + ".txt"], env=dict(os.environ, XAAC_CAPTURE_FILE=tmp), check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    recs = c.read_records(tmp)
    os.remove(tmp)
    return recs


def main():
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_harness.so"))
    ref.ref_sbr_dec_eld.restype = ctypes.c_int
    ref.ref_sbr_dec_eld.argtypes = [ctypes.c_void_p] * 3 + [P16, ctypes.c_int, P16, ctypes.c_int, P32]
    rng = np.random.default_rng(20260930)
    H, F, RET, CRC, ST0, CH, N = [], [], [], [], [], [], []
    chain = 0
    for name in STREAMS:
        recs = captured(name)
        assert recs and all(r["aot"] == 39 and not r["low_pow"] for r in recs), name
        for p in range(PASSES):
            for ch in range(2):                     # the streams are stereo: a channel's calls alternate
                mine = [r for i, r in enumerate(recs) if i % 2 == ch]
                st = c.eld_state_from(mine[0]["st0"])
                ST0.append(np.frombuffer(bytes(st), np.uint8).copy())
                n = mine[0]["header"].num_time_slots
                for s, r in enumerate(mine):
                    h = c.Header.from_buffer_copy(bytes(r["header"]))
                    f = c.Frame.from_buffer_copy(bytes(r["frame"]))
                    if p and s % 5 != 4:
                        fuzz_sbr(rng, h, f, True)
                    pin = np.ascontiguousarray(chain_pcm(3, chain, s)[:32 * n])
                    po = np.zeros(64 * n, np.int16)
                    hand = np.zeros(16 * 128, np.int32)
                    ret = ref.ref_sbr_dec_eld(ctypes.byref(h), ctypes.byref(f), ctypes.byref(st), pin.ctypes.data_as(P16), 1,
                                              po.ctypes.data_as(P16), 1, hand.ctypes.data_as(P32))
                    H.append(np.frombuffer(bytes(h), np.uint8).copy()); F.append(np.frombuffer(bytes(f), np.uint8).copy())
                    RET.append(ret); CRC.append((crc(po.tobytes()), crc(st), crc(hand[:128 * n].tobytes()))); CH.append(chain)
                N.append(n)
                chain += 1
    d = dict(header=np.stack(H), frame=np.stack(F), ret=np.array(RET, np.int32), crc=np.array(CRC, np.uint32),
             step_chain=np.array(CH, np.int32), st0=np.stack(ST0), n_slots=np.array(N, np.int32))
    dst = os.path.join(ROOT, "tests", "golden", "sbr_eld_chains.npz")
    np.savez_compressed(dst, **d)
    print(dst, os.path.getsize(dst), "bytes;", len(RET), "steps in", chain, "chains;", int(np.count_nonzero(d["ret"])), "steps returned an error;",
          "slots per frame:", sorted(set(N)))


if __name__ == "__main__":
    main()
