#!/usr/bin/env python3
"""Build tests/golden/sbr_chains.npz: >= 1000 steps each of the REAL ixheaacd_sbr_dec in low-power mode (HE-AACv1) and
in HQ mode with parametric stereo (HE-AACv2), as CHAINS with the state carried from call to call by the reference
itself (oracle/ref_sbr_adapter.c drives the compiled reference, oracle/_ref/libref_harness.so).

A chain starts from the state the reference had at the start of a captured stream (tools/make_test_streams.py +
oracle/_ref/xaacdec_capture) and walks that stream's captured frames in order from a random offset, each with
reference-side fuzz: inverse-filter modes, limiter gains, interpolation, gain smoothing, added harmonics, envelope
energies (exponent nudges), and for HE-AACv2 the fine / coarse IID quantiser, 1-5 PS envelopes with random borders,
random IID / ICC indices.  The grids (FIXFIX / FIXVAR / VARFIX / VARVAR, 1-5 envelopes, transients) are the captured
streams' own.  The core PCM of a step is synthetic and is NOT stored: tests regenerate it from (chain, step) with
chain_pcm() below -- a counter-based generator written out in integer arithmetic, no library RNG.  Stored per step:
header, frame, PS frame, the reference's return code, and CRC32s of its PCM output and of its state(s) after the call
(the full states at the chain's start are stored once).  Data only; runs only where /root/reference exists."""
import ctypes
import glob
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sbr_capture as c  # noqa: E402

P16 = ctypes.POINTER(ctypes.c_int16)
STEPS = 44
AMPS = (30000, 3000, 12000, 200, 800, 32767)


def chain_pcm(kind, chain, step):
    """1024 core samples of (kind 0 = LP / 1 = HQ, chain, step): splitmix64 of a counter, scaled to the step's level"""
    base = np.uint64((kind << 40) | (chain << 20) | step) * np.uint64(1024)
    z = (base + np.arange(1024, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    v = (z >> np.uint64(40)).astype(np.int64) - (1 << 23)             # 24-bit signed, uniform
    amp = AMPS[(chain + step) % len(AMPS)]
    return ((v * amp) >> 23).astype(np.int16)


def crc(b):
    return zlib.crc32(bytes(b)) & 0xffffffff


def fuzz_sbr(rng, h, f, hq):
    for k in range(h.num_if_bands):
        f.sbr_invf_mode[k] = int(rng.integers(0, 4))
    h.limiter_gains = int(rng.integers(0, 4))
    h.interpol_freq = int(rng.integers(0, 2))
    if hq:
        h.smoothing_mode = int(rng.integers(0, 2))
    k = int(rng.integers(0, 4))
    if k == 0:
        for b in range(h.num_sf_bands[1]):
            f.add_harmonics[b] = int(rng.integers(0, 4) == 0)
    elif k == 1:
        for b in range(h.num_sf_bands[1]):
            f.add_harmonics[b] = 0
    if rng.integers(0, 3) == 0:      # envelope energies: nudge the exponents (low 6 bits of the packed value)
        n = sum(h.num_sf_bands[f.freq_res[e]] for e in range(f.num_env))
        for i in range(n):
            v = f.int_env_sf_arr[i]
            e = int(np.clip((v & 63) + rng.integers(-2, 3), 0, 40))
            f.int_env_sf_arr[i] = (v & ~63) | e


def fuzz_ps(rng, pf):
    pf.iid_quant = int(rng.integers(0, 2))
    nenv = int(rng.integers(1, 6))
    borders = [0] + sorted(rng.choice(np.arange(1, 32), nenv - 1, replace=False).tolist()) + [32]
    for e in range(7):
        pf.border_position[e] = borders[e] if e < len(borders) else 0
    lim = 15 if pf.iid_quant else 7
    for e in range(nenv):
        for b in range(34):
            pf.iid_par_table[e][b] = int(rng.integers(-lim, lim + 1))
            pf.icc_par_table[e][b] = int(rng.integers(0, 8))


def build(ref, rng, caps, hq, n_chains):
    kind = 1 if hq else 0
    H, F, PF, RET, C_PCM, C_ST, C_PS, ST0, PS0 = [], [], [], [], [], [], [], [], []
    for ci in range(n_chains):
        recs = caps[ci % len(caps)]
        if not hq:   # one channel of the stream: every second call belongs to it in a stereo stream
            ch = ci // len(caps) % 2
            recs = [r for i, r in enumerate(recs) if i % 2 == ch]   # (the captured HE-AACv1 streams are stereo: calls alternate)
        st = c.State.from_buffer_copy(bytes(recs[0]["st0"]))
        ps = c.PsState.from_buffer_copy(bytes(recs[0]["ps0"])) if hq else None
        ST0.append(np.frombuffer(bytes(st), np.uint8).copy())
        if hq:
            PS0.append(np.frombuffer(bytes(ps), np.uint8).copy())
        off = int(rng.integers(0, max(1, len(recs) - STEPS)))
        hs, fs, pfs, rets, cp, cs, cps = [], [], [], [], [], [], []
        for s in range(STEPS):
            r = recs[(off + s) % len(recs)] if s else recs[0]      # step 0 = the stream's own start-up frame
            h = c.Header.from_buffer_copy(bytes(r["header"]))
            f = c.Frame.from_buffer_copy(bytes(r["frame"]))
            if s % 5 != 4:
                fuzz_sbr(rng, h, f, hq)
            pin = np.ascontiguousarray(chain_pcm(kind, ci, s))
            if hq:
                pf = c.PsFrame.from_buffer_copy(bytes(r["ps_frame"]))
                if s % 3 != 2:
                    fuzz_ps(rng, pf)
                po = np.zeros(4096, np.int16)
                ret = ref.ref_sbr_dec_hq(ctypes.byref(h), ctypes.byref(f), ctypes.byref(st), ctypes.byref(pf), ctypes.byref(ps),
                                         pin.ctypes.data_as(P16), 1, po.ctypes.data_as(P16), 2)
                pfs.append(np.frombuffer(bytes(pf), np.uint8).copy())
                cps.append(crc(ps))
            else:
                po = np.zeros(2048, np.int16)
                ret = ref.ref_sbr_dec_lp(ctypes.byref(h), ctypes.byref(f), ctypes.byref(st), pin.ctypes.data_as(P16), 1,
                                         po.ctypes.data_as(P16), 1)
            assert ret == 0, (ci, s, ret)
            hs.append(np.frombuffer(bytes(h), np.uint8).copy()); fs.append(np.frombuffer(bytes(f), np.uint8).copy())
            rets.append(ret); cp.append(crc(po.tobytes())); cs.append(crc(st))
        H.append(hs); F.append(fs); RET.append(rets); C_PCM.append(cp); C_ST.append(cs)
        if hq:
            PF.append(pfs); C_PS.append(cps)
    pre = "hq_" if hq else "lp_"
    d = {pre + "header": np.array(H), pre + "frame": np.array(F), pre + "ret": np.array(RET, np.int32),
         pre + "crc_pcm": np.array(C_PCM, np.uint32), pre + "crc_state": np.array(C_ST, np.uint32), pre + "st0": np.array(ST0)}
    if hq:
        d.update({"hq_ps_frame": np.array(PF), "hq_crc_ps": np.array(C_PS, np.uint32), "hq_ps0": np.array(PS0)})
    return d


def main(stream_dir):
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_harness.so"))
    rng = np.random.default_rng(2030)
    lp = [c.read_records(p) for p in sorted(glob.glob(os.path.join(stream_dir, "*aot5*.cap")))]
    hq = [c.read_records(p) for p in sorted(glob.glob(os.path.join(stream_dir, "*aot29*.cap")))]
    lp = [[r for r in recs if r["enh"] == 0] for recs in lp]
    hq = [[r for r in recs if r["enh"] == 0] for recs in hq]
    d = build(ref, rng, lp, False, 24)
    d.update(build(ref, rng, hq, True, 24))
    dst = os.path.join(ROOT, "tests", "golden", "sbr_chains.npz")
    np.savez_compressed(dst, **d)
    print(dst, os.path.getsize(dst), "bytes;", d["lp_ret"].size, "LP steps,", d["hq_ret"].size, "HQ+PS steps")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/xaac_streams")
