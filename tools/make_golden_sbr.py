#!/usr/bin/env python3
"""Build tests/golden/sbr_lp_records.bin.gz: ~70 calls of the REAL ixheaacd_sbr_dec (low-power SBR) in the
record format of oracle/ref_capture.c -- inputs, state before, state after, output PCM -- for the box that has
no reference.  Half are calls captured while the reference decoded HE-AACv1 streams (tools/make_test_streams.py),
chosen to cover start-up, 1-4 envelopes, transients, sinusoidal coding and three frequency tables; half are the
same frames with fuzzed side info (inverse-filter modes, limiter gains, interpolation, added harmonics) pushed
through the reference by oracle/ref_sbr_adapter.c."""
import ctypes
import glob
import gzip
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sbr_capture as c  # noqa: E402

P16 = ctypes.POINTER(ctypes.c_int16)


def pack(meta, h, f, st0, pcm_in, st1, pcm_out):
    return (np.asarray(meta, np.int32).tobytes() + bytes(h) + bytes(f) + bytes(st0) + pcm_in.astype(np.int16).tobytes() +
            bytes(st1) + pcm_out.astype(np.int16).tobytes())


def main(stream_dir):
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_harness.so"))
    rng = np.random.default_rng(2026)
    out = []
    for path in sorted(glob.glob(os.path.join(stream_dir, "*aot5*.cap"))):
        recs = c.read_records(path)
        seen = {}
        for i, r in enumerate(recs):
            key = (r["frame"].num_env, r["frame"].transient_env >= 0, sum(r["frame"].add_harmonics) > 0)
            if key not in seen and i > 8:
                seen[key] = i
        # start-up calls of both channels + the richest frame kinds (most envelopes first)
        picks = [0] + [seen[k] for k in sorted(seen, reverse=True)[:3]]
        for i in picks:
            r = recs[i]
            meta = [0x58414331, r["call"], 1, 1, r["aot"], 0, r["ret"], 0]
            out.append(pack(meta, r["header"], r["frame"], r["st0"], r["pcm_in"], r["st1"], r["pcm_out"]))
            # fuzzed twin through the reference adapter
            h = c.Header.from_buffer_copy(bytes(r["header"]))
            f = c.Frame.from_buffer_copy(bytes(r["frame"]))
            for k in range(h.num_if_bands):
                f.sbr_invf_mode[k] = int(rng.integers(0, 4))
            h.limiter_gains = int(rng.integers(0, 4))
            h.interpol_freq = int(rng.integers(0, 2))
            if rng.integers(0, 2):
                for k in range(h.num_sf_bands[1]):
                    f.add_harmonics[k] = int(rng.integers(0, 3) == 0)
            st = c.State.from_buffer_copy(bytes(r["st0"]))
            for k in range(h.num_if_bands):
                st.prev_invf_mode[k] = int(rng.integers(0, 4))
                st.bw_array_prev[k] = int(rng.integers(0, 0x7f800000))
            st0 = c.State.from_buffer_copy(bytes(st))
            pin = np.ascontiguousarray(r["pcm_in"])
            po = np.zeros((2, 2048), np.int16)
            ret = ref.ref_sbr_dec_lp(ctypes.byref(h), ctypes.byref(f), ctypes.byref(st), pin.ctypes.data_as(P16), 1,
                                     po[0].ctypes.data_as(P16), 1)
            meta = [0x58414331, r["call"], 1, 1, r["aot"], 0, ret, 1]
            out.append(pack(meta, h, f, st0, pin, st, po))
    dst = os.path.join(ROOT, "tests", "golden", "sbr_lp_records.bin.gz")
    with gzip.open(dst, "wb", 9) as g:
        g.write(b"".join(out))
    print(dst, os.path.getsize(dst), "bytes,", len(out), "records")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/xaac_streams")
