"""Down-sampled SBR (32-channel synthesis bank) test cases built from the committed golden records: the record's side
info and state with a fresh synthesis bank (a 64-channel ring position means nothing to the 640-sample ring), three
frames chained with seeded core PCM.  Test infrastructure."""
import ctypes
import os

import numpy as np

import sbr_capture as cap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P16 = ctypes.POINTER(ctypes.c_int16)
LP_GOLD = os.path.join(ROOT, "tests", "golden", "sbr_lp_records.bin.gz")
HQ_GOLD = os.path.join(ROOT, "tests", "golden", "sbr_hq_ps_records.bin.gz")
FRAMES = 3


def fresh_bank(st0):
    st = cap.State.from_buffer_copy(bytes(st0))
    ctypes.memset(ctypes.addressof(st) + cap.State.syn_ring.offset, 0, cap.State.syn_ring.size)
    st.syn_drc_offset = 0
    st.syn_phase = 0
    return st


def cases(low_pow, limit=16):
    recs = cap.read_records(LP_GOLD if low_pow else HQ_GOLD)
    step = max(1, len(recs) // limit)
    recs = recs[::step][:limit]
    if not low_pow:   # HE-AAC mono in HQ mode: the HE-AACv2 records without the PS tool
        for r in recs:
            h = cap.Header.from_buffer_copy(bytes(r["header"]))
            h.channel_mode = 1
            r["header"] = h
    return recs


def core_pcm(seed, frame):
    rng = np.random.default_rng(1000 * seed + frame)
    return np.ascontiguousarray((rng.standard_normal(1024) * 3000 * (1 + frame)).clip(-32768, 32767).astype(np.int16))


def run_chain(call, low_pow, recs):
    """call(h, f, st, pcm_in int16[1024], out int16[1024]) -> rc; returns (outputs [n, FRAMES, 1024], rcs, final states)"""
    outs, rcs, states = [], [], []
    for i, r in enumerate(recs):
        st = fresh_bank(r["st0"])
        for k in range(FRAMES):
            out = np.zeros(1024, np.int16)
            rcs.append(call(r["header"], r["frame"], st, core_pcm(i, k), out))
            outs.append(out)
        states.append(st)
    return np.stack(outs).reshape(len(recs), FRAMES, 1024), rcs, states
