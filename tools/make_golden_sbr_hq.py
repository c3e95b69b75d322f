#!/usr/bin/env python3
"""Build tests/golden/sbr_hq_ps_records.bin.gz: ~50 calls of the REAL ixheaacd_sbr_dec in HQ mode with parametric
stereo (HE-AACv2) in the record format of oracle/ref_capture.c -- inputs, SBR and PS state before / after, both
output channels -- for the box that has no reference.  Half are calls captured while the reference decoded
HE-AACv2 streams (tools/make_test_streams.py), chosen to cover start-up, 1-5 envelopes and sinusoidal coding; half
are the same frames with fuzzed side info (inverse-filter modes -> the complex LPC filter, gain smoothing, limiter
gains, energies per scale-factor band, added harmonics, the fine IID quantiser, several PS envelopes, random IID /
ICC indices) pushed through the reference by oracle/ref_sbr_adapter.c."""
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


def pack(meta, h, f, st0, pcm_in, st1, pcm_out, pf, ps0, ps1):
    return (np.asarray(meta, np.int32).tobytes() + bytes(h) + bytes(f) + bytes(st0) + pcm_in.astype(np.int16).tobytes() +
            bytes(st1) + pcm_out.astype(np.int16).tobytes() + bytes(pf) + bytes(ps0) + bytes(ps1))


def fuzz(rng, h, f, pf):
    for k in range(h.num_if_bands):
        f.sbr_invf_mode[k] = int(rng.integers(0, 4))
    h.limiter_gains = int(rng.integers(0, 4))
    h.interpol_freq = int(rng.integers(0, 2))
    h.smoothing_mode = int(rng.integers(0, 2))
    if rng.integers(0, 2):
        for k in range(h.num_sf_bands[1]):
            f.add_harmonics[k] = int(rng.integers(0, 3) == 0)
    pf.iid_quant = int(rng.integers(0, 2))
    nenv = int(rng.integers(1, 5))
    borders = [0] + sorted(rng.choice(np.arange(1, 32), nenv - 1, replace=False).tolist()) + [32]
    for e in range(7):
        pf.border_position[e] = borders[e] if e < len(borders) else 0
    lim = 15 if pf.iid_quant else 7
    for e in range(nenv):
        for b in range(20):
            pf.iid_par_table[e][b] = int(rng.integers(-lim, lim + 1))
            pf.icc_par_table[e][b] = int(rng.integers(0, 8))


def main(stream_dir):
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_harness.so"))
    rng = np.random.default_rng(2029)
    out = []
    for path in sorted(glob.glob(os.path.join(stream_dir, "*aot29*.cap"))):
        recs = c.read_records(path)
        seen = {}
        for i, r in enumerate(recs):
            key = (r["frame"].num_env, sum(r["frame"].add_harmonics) > 0)
            if key not in seen and i > 8:
                seen[key] = i
        picks = [0] + [seen[k] for k in sorted(seen, reverse=True)[:3]]
        for i in picks:
            r = recs[i]
            meta = [0x58414331, r["call"], 0, 2, r["aot"], 1, r["ret"], 0]
            out.append(pack(meta, r["header"], r["frame"], r["st0"], r["pcm_in"], r["st1"], r["pcm_out"], r["ps_frame"],
                            r["ps0"], r["ps1"]))
            h = c.Header.from_buffer_copy(bytes(r["header"]))
            f = c.Frame.from_buffer_copy(bytes(r["frame"]))
            pf = c.PsFrame.from_buffer_copy(bytes(r["ps_frame"]))
            fuzz(rng, h, f, pf)
            st = c.State.from_buffer_copy(bytes(r["st0"]))
            ps = c.PsState.from_buffer_copy(bytes(r["ps0"]))
            for k in range(h.num_if_bands):
                st.prev_invf_mode[k] = int(rng.integers(0, 4))
                st.bw_array_prev[k] = int(rng.integers(0, 0x7f800000))
            st0 = c.State.from_buffer_copy(bytes(st))
            ps0 = c.PsState.from_buffer_copy(bytes(ps))
            pin = np.ascontiguousarray(r["pcm_in"])
            po = np.zeros(4096, np.int16)
            ret = ref.ref_sbr_dec_hq(ctypes.byref(h), ctypes.byref(f), ctypes.byref(st), ctypes.byref(pf), ctypes.byref(ps),
                                     pin.ctypes.data_as(P16), 1, po.ctypes.data_as(P16), 2)
            meta = [0x58414331, r["call"], 0, 2, r["aot"], 1, ret, 1]
            out.append(pack(meta, h, f, st0, pin, st, np.stack([po[0::2], po[1::2]]), pf, ps0, ps))
    dst = os.path.join(ROOT, "tests", "golden", "sbr_hq_ps_records.bin.gz")
    with gzip.open(dst, "wb", 9) as g:
        g.write(b"".join(out))
    print(dst, os.path.getsize(dst), "bytes,", len(out), "records")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/xaac_streams")
