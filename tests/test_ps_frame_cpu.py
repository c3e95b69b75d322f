"""The frame-at-once arrangement of the parametric-stereo tool (libxaac_amd/csrc/sbr_ps_frame.h, what the GPU kernel
runs) against the slot loop that restates the reference (libxaac_amd/csrc/sbr_ps.h, what the oracle runs and what is
pinned to the compiled reference): both compiled for the host, same inputs -> same PCM, SBR state and PS state, on the
reference's captured HE-AACv2 frames and on chains of fuzzed side info including borders no parser produces
(unsorted, first border not at slot 0, repeated) and band limits that change from frame to frame."""
import ctypes
import os

import numpy as np

import sbr_capture as cap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P16 = ctypes.POINTER(ctypes.c_int16)
GOLDEN = os.path.join(ROOT, "tests", "golden", "sbr_hq_ps_records.bin.gz")


def both(oracle, h, f, st, pf, ps, pcm):
    outs = []
    for name in ("xo_sbr_dec_hq", "xo_sbr_dec_hq_phased"):
        s, p = cap.State.from_buffer_copy(bytes(st)), cap.PsState.from_buffer_copy(bytes(ps))
        out = np.zeros(4096, np.int16)
        rc = getattr(oracle.lib, name)(ctypes.byref(h), ctypes.byref(f), ctypes.byref(s), ctypes.byref(pf), ctypes.byref(p),
                                       pcm.ctypes.data_as(P16), 1, out.ctypes.data_as(P16), 2)
        outs.append((rc, out, s, p))
    return outs


def same(a, b, tag):
    assert a[0] == b[0], tag
    assert np.array_equal(a[1], b[1]), (tag, "pcm", int(np.sum(a[1] != b[1])))
    assert not cap.diff_state(a[2], b[2]), (tag, cap.diff_state(a[2], b[2])[:3])
    assert not cap.diff_state(a[3], b[3]), (tag, cap.diff_state(a[3], b[3])[:3])


def test_reference_records(oracle):
    for i, r in enumerate(cap.read_records(GOLDEN)):
        a, b = both(oracle, r["header"], r["frame"], r["st0"], r["ps_frame"], r["ps0"], r["pcm_in"])
        same(a, b, i)
        assert np.array_equal(b[1][0::2], r["pcm_out"][0]) and np.array_equal(b[1][1::2], r["pcm_out"][1])
        assert not cap.diff_state(b[3], r["ps1"])


def _fuzz_ps(rng, pf, wild):
    pf.iid_quant = int(rng.integers(0, 2))
    nenv = int(rng.integers(1, 6))
    if wild == 0:      # what a parser makes: 0 = b0 < b1 < ... <= 32
        borders = [0] + sorted(rng.choice(np.arange(1, 32), nenv - 1, replace=False).tolist()) + [32]
    elif wild == 1:    # first border not at slot 0 (or never reached)
        borders = sorted(rng.choice(np.arange(1, 40), nenv, replace=False).tolist()) + [32]
    else:              # anything, repeated and unsorted
        borders = rng.integers(-3, 36, nenv + 1).tolist()
    for e in range(7):
        pf.border_position[e] = int(borders[e]) if e < len(borders) else int(rng.integers(0, 33))
    lim = 15 if pf.iid_quant else 7
    for e in range(7):
        for b in range(34):
            pf.iid_par_table[e][b] = int(rng.integers(-lim, lim + 1))
            pf.icc_par_table[e][b] = int(rng.integers(0, 8))


def test_fuzzed_chains(oracle):
    recs = cap.read_records(GOLDEN)
    rng = np.random.default_rng(2027)
    for i, r in enumerate(recs):
        st, ps = cap.State.from_buffer_copy(bytes(r["st0"])), cap.PsState.from_buffer_copy(bytes(r["ps0"]))
        for step in range(8):
            f = cap.Frame.from_buffer_copy(bytes(r["frame"]))
            pf = cap.PsFrame.from_buffer_copy(bytes(r["ps_frame"]))
            _fuzz_ps(rng, pf, (i + step) % 3)
            if step % 4 == 3:      # the band limit moves: delay lines of newly active bands are cleared at border 0
                f.max_qmf_subband_aac = int(np.clip(f.max_qmf_subband_aac + rng.integers(-6, 7), r["header"].sub_band_start, 32))
            if step == 5:
                st.syn_usb = int(rng.integers(8, 30))    # an upper limit below the all-pass bands
            amp = [30000, 3000, 200, 12][step % 4]
            pcm = rng.integers(-amp, amp + 1, 1024).astype(np.int16)
            a, b = both(oracle, r["header"], f, st, pf, ps, pcm)
            same(a, b, (i, step))
            st, ps = a[2], a[3]
