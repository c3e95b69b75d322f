#!/usr/bin/env python3
"""Build tests/golden/esbr_chains.npz: >= 1000 steps of the REAL ixheaacd_sbr_dec taking its eSBR branch ("Path A",
sbr_dec.c:816-1009: analysis with 32-bit rings, QMF harmonic transposer, float HF generator + envelope adjuster, float
parametric stereo, synthesis) as CHAINS whose state the reference carries itself.

oracle/_ref/xaacdec_capture (oracle/ref_capture.c: esbr_chain_call) decodes the committed HE-AAC streams several times;
at every eSBR call it fuzzes the reference's own live side info within what a bitstream can say (limiter gains / bands,
pre-flattening, interpolation, smoothing, inverse-filter modes, added harmonics, inter-TES, harmonic patching with and without a pitch,
resets, PS quantiser / 1-4 envelopes / IID / ICC indices), replaces the core input by a counter-based synthetic frame
(chain_core() below regenerates it: nothing is stored), runs the real function and writes the step in the boundary
formats of include/xaac_esbr.h: side info, return code, CRC32s of out / out_r and of the three states after the call.
A chain = one channel of one decoder run.  Data only; runs only where /root/reference exists."""
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden_sbr_chains import chain_pcm  # noqa: E402

# (harm_aot5_48k.aac carries ENHSBR elements of its own: harmonic patching and pre-flattened LPP frames; the fuzz adds them
# to the other streams, with inter-TES)
STREAMS = ("mix_aot5_48k", "mono_aot5_32k", "mix_aot29_32k", "harm_aot5_48k")
PASSES = 10           # decoder runs per stream; pass 0 is unfuzzed
# `python tools/make_golden_esbr_chains.py usac` -> tests/golden/esbr_usac_chains.npz: the same chains for USAC channels
# (tests/golden/streams_usac, raw access units + the encoder's frame-size list: xaacdec -mp4:1 -imeta:) -- stereo 2:1 eSBR
# without a harmonic transposer (codec_x_delay 0: xaac_esbr.h XAAC_ESBR_NO_X_DELAY), with one (sbr_patching_mode 0 frames),
# a switched FD / LPD core (the ORIG_SBR frames of it)
USAC_STREAMS = ("u21", "u21harm", "u21sw", "m21swpvc", "m21tdpvc")
# `python tools/make_golden_esbr_chains.py ratios` -> tests/golden/esbr_ratio_chains.npz: USAC channels at the other two SBR
# ratios -- 8:3 (768-sample core frames through the 24-channel bank) and 4:1 (the 16-channel bank, 64 slots) -- plain, with
# PVC frames (the reference encoder makes no harmonic SBR at these ratios); "chain_ratio" says which (xaac_esbr.h: XAAC_ESBR_RATIO_*)
RATIO_STREAMS = ("u83", "m83swpvc", "u41", "m41swpvc", "m41tdpvc")
USAC_PASSES = 6


def chain_core(run, chain, step):
    """the float core frame of (run, chain-in-run, step): the integers of chain_pcm(kind 2) as floats"""
    return chain_pcm(2, run * 32 + chain, step).astype(np.float32)


def sizes():
    import ctypes
    import esbr_structs as es
    import hbe_structs as hs
    import sbr_capture as c
    return dict(hd=ctypes.sizeof(c.Header), fr=ctypes.sizeof(c.Frame), sd=ctypes.sizeof(es.EsbrSide), psf=ctypes.sizeof(c.PsFrame),
                est=ctypes.sizeof(es.EsbrState), hbs=ctypes.sizeof(hs.HbeState), eps=ctypes.sizeof(es.EsbrPsState),
                pvs=ctypes.sizeof(es.EsbrPvcSide), pvst=ctypes.sizeof(es.EsbrPvcState))


def parse(path, run, z, pvc=False):
    """pvc: a $XAAC_ESBR_CHAIN_PVC run -- the PVC state behind a chain's first states, the PVC side info behind every step's
    PS frame, a sixth CRC (the PVC state after the call)"""
    b = open(path, "rb").read()
    o, recs = 0, []
    while o < len(b):
        magic, chain, step, eps, first, apply = struct.unpack_from("<6i", b, o)
        assert magic == 0x58414332, hex(magic)
        o += 24
        eps, ratio = eps & 0xff, eps >> 8
        r = dict(run=run, chain=chain, step=step, eps=eps, first=first, apply=apply, ratio=ratio)
        if first:
            r["est0"] = b[o:o + z["est"]]; o += z["est"]
            r["hbs0"] = b[o:o + z["hbs"]]; o += z["hbs"]
            if eps:
                r["eps0"] = b[o:o + z["eps"]]; o += z["eps"]
            if pvc:
                r["pvst0"] = b[o:o + z["pvst"]]; o += z["pvst"]
        for k in ("hd", "fr", "sd", "psf") + (("pvs",) if pvc else ()):
            r[k] = b[o:o + z[k]]; o += z[k]
        r["ret"], *crcs = struct.unpack_from("<i6I" if pvc else "<i5I", b, o)
        r["crc"] = crcs
        o += 28 if pvc else 24
        recs.append(r)
    return recs


def main(usac=False, ratios=False):
    usac = usac or ratios
    z = sizes()
    cap = os.path.join(ROOT, "oracle", "_ref", "xaacdec_capture")
    recs = []
    run = 0
    for p in range(USAC_PASSES if usac else PASSES):
        for s in (RATIO_STREAMS if ratios else USAC_STREAMS if usac else STREAMS):
            tmp = "/tmp/xaac_esbr_chain_%d.bin" % run
            env = dict(os.environ, XAAC_ESBR_CHAIN_FILE=tmp, XAAC_ESBR_CHAIN_SEED=str(0 if p == 0 else 100 * p + run),
                       XAAC_ESBR_CHAIN_RUN=str(run))
            if usac:
                env["XAAC_ESBR_CHAIN_PVC"] = "1"
            src = os.path.join(ROOT, "tests", "golden", "streams_usac" if usac else "streams", s + ".aac")
            args = ["-ifile:" + src, "-ofile:/tmp/xaac_esbr_chain.wav"] + (["-mp4:1", "-imeta:" + src[:-4] + ".txt"] if usac else [])
            subprocess.run([cap] + args, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            recs += parse(tmp, run, z, pvc=usac)
            os.remove(tmp)
            run += 1
    chains = sorted({(r["run"], r["chain"]) for r in recs})
    cidx = {c: i for i, c in enumerate(chains)}
    n, nc = len(recs), len(chains)
    u8 = lambda x, size: np.frombuffer(x, np.uint8) if x is not None else np.zeros(size, np.uint8)
    d = {
        "chain_run": np.array([c[0] for c in chains], np.int32), "chain_id": np.array([c[1] for c in chains], np.int32),
        "chain_ps": np.zeros(nc, np.int32), "chain_len": np.zeros(nc, np.int32),
        "est0": np.zeros((nc, z["est"]), np.uint8), "hbs0": np.zeros((nc, z["hbs"]), np.uint8), "eps0": np.zeros((nc, z["eps"]), np.uint8),
        "step_chain": np.array([cidx[(r["run"], r["chain"])] for r in recs], np.int32),
        "step_idx": np.array([r["step"] for r in recs], np.int32), "apply": np.array([r["apply"] for r in recs], np.int32),
        "ret": np.array([r["ret"] for r in recs], np.int32), "crc": np.array([r["crc"] for r in recs], np.uint32),
        "header": np.stack([u8(r["hd"], 0) for r in recs]), "frame": np.stack([u8(r["fr"], 0) for r in recs]),
        "side": np.stack([u8(r["sd"], 0) for r in recs]), "ps_frame": np.stack([u8(r["psf"], 0) for r in recs]),
    }
    if usac:
        d["pvc_side"] = np.stack([u8(r["pvs"], 0) for r in recs])
        d["chain_ratio"] = np.zeros(nc, np.int32)
        for r in recs:
            d["chain_ratio"][cidx[(r["run"], r["chain"])]] = r["ratio"]
        d["pvst0"] = np.zeros((nc, z["pvst"]), np.uint8)
        for r in recs:
            if r["first"]:
                d["pvst0"][cidx[(r["run"], r["chain"])]] = u8(r["pvst0"], 0)
    for r in recs:
        i = cidx[(r["run"], r["chain"])]
        d["chain_len"][i] += 1
        d["chain_ps"][i] = r["eps"]
        if r["first"]:
            d["est0"][i] = u8(r["est0"], 0)
            d["hbs0"][i] = u8(r["hbs0"], 0)
            if r["eps"]:
                d["eps0"][i] = u8(r["eps0"], 0)
    dst = os.path.join(ROOT, "tests", "golden", "esbr_ratio_chains.npz" if ratios else "esbr_usac_chains.npz" if usac else "esbr_chains.npz")
    np.savez_compressed(dst, **d)
    from esbr_structs import EsbrSide
    harm = sum(1 for r in recs if r["apply"] and np.frombuffer(r["sd"], np.int16)[EsbrSide.harmonic_sbr.offset // 2] != 0)
    flat = sum(1 for r in recs if r["apply"] and np.frombuffer(r["sd"], np.int16)[EsbrSide.harmonic_sbr.offset // 2] == 2)
    print(dst, os.path.getsize(dst), "bytes;", n, "steps in", nc, "chains;", int(d["ret"].astype(bool).sum()), "steps returned an error;",
          int(d["apply"].sum()), "processed;", harm, "with harmonic patching or the pre-flattening flag;", flat, "with pre-flattened LPP patches")


if __name__ == "__main__":
    main(usac=len(sys.argv) > 1 and sys.argv[1] == "usac", ratios=len(sys.argv) > 1 and sys.argv[1] == "ratios")
