"""The envelope adjuster's two-envelopes-per-pass arrangement (libxaac_amd/csrc/sbr_core.h: "two envelopes side by side",
what the HQ core kernel and the oracle run for regular frames) against the one-envelope chain (the literal restatement of
ixheaacd_calc_sbrenvelope's loop, env_calc.c:692; oracle/oracle_sbr_seq.cpp compiles the oracle with XS_NO_ENV_PAIRS), both
on the host, same inputs -> same PCM, SBR state and PS state
  * on the reference's captured HE-AACv2 frames (and against the reference's own outputs),
  * on the 24 reference-made chains of tests/golden/sbr_chains.npz walked by the one-envelope build alone (the paired build
    walks them in tests/test_sbr_chains.py),
  * on chains with fuzzed frame side info: envelope borders anywhere in 0..19 (unsorted, empty, behind slot 32: these pairs are
    refused and fall back), frequency resolutions that differ inside a pair, noise-floor rows that switch between the two
    envelopes, transient envelopes, inverse-filter modes, harmonics, limiter gains, band limits that move.
CPU only."""
import ctypes
import os
import sys
import zlib

import numpy as np

import sbr_capture as cap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden_sbr_chains import chain_pcm  # noqa: E402

P16 = ctypes.POINTER(ctypes.c_int16)
GOLDEN = os.path.join(ROOT, "tests", "golden", "sbr_hq_ps_records.bin.gz")


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def both(oracle, h, f, st, pf, ps, pcm):
    outs = []
    for name in ("xo_sbr_dec_hq", "xo_sbr_dec_hq_seq"):
        s, p = cap.State.from_buffer_copy(bytes(st)), cap.PsState.from_buffer_copy(bytes(ps))
        out = np.zeros(4096, np.int16)
        rc = getattr(oracle.lib, name)(ctypes.byref(h), ctypes.byref(f), ctypes.byref(s), ctypes.byref(pf), ctypes.byref(p),
                                       pcm.ctypes.data_as(P16), 1, out.ctypes.data_as(P16), 2)
        outs.append((rc, out, s, p))
    return outs


def same(a, b, tag):
    assert a[0] == b[0], (tag, a[0], b[0])
    if a[0] != 0:
        return
    assert not cap.diff_state(a[2], b[2]), (tag, cap.diff_state(a[2], b[2])[:4])
    assert np.array_equal(a[1], b[1]), (tag, "pcm", int(np.sum(a[1] != b[1])))
    assert not cap.diff_state(a[3], b[3]), (tag, cap.diff_state(a[3], b[3])[:3])


def test_reference_records(oracle):
    n = 0
    for i, r in enumerate(cap.read_records(GOLDEN)):
        a, b = both(oracle, r["header"], r["frame"], r["st0"], r["ps_frame"], r["ps0"], r["pcm_in"])
        same(a, b, i)
        assert np.array_equal(b[1][0::2], r["pcm_out"][0]) and np.array_equal(b[1][1::2], r["pcm_out"][1])
        assert not cap.diff_state(b[2], r["st1"]) and not cap.diff_state(b[3], r["ps1"])
        n += 1
    assert n >= 24


def test_reference_made_chains(oracle):
    """the one-envelope build alone walks the reference's chains (its own state carried) and reproduces every CRC"""
    ch = np.load(os.path.join(ROOT, "tests", "golden", "sbr_chains.npz"))
    hdr, frm, ret = ch["hq_header"], ch["hq_frame"], ch["hq_ret"]
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for c in range(ret.shape[0]):
        st = np.ascontiguousarray(ch["hq_st0"][c]).copy()
        ps = np.ascontiguousarray(ch["hq_ps0"][c]).copy()
        for s in range(ret.shape[1]):
            pin = np.ascontiguousarray(chain_pcm(1, c, s))
            h, f, pf = (np.ascontiguousarray(ch[k][c, s]) for k in ("hq_header", "hq_frame", "hq_ps_frame"))
            out = np.zeros(4096, np.int16)
            rc = oracle.lib.xo_sbr_dec_hq_seq(vp(h), vp(f), vp(st), vp(pf), vp(ps), pin.ctypes.data_as(P16), 1,
                                              out.ctypes.data_as(P16), 2)
            assert rc == ret[c, s], (c, s)
            assert crc(out) == ch["hq_crc_pcm"][c, s], ("pcm", c, s)
            assert crc(st) == ch["hq_crc_state"][c, s], ("state", c, s)
            assert crc(ps) == ch["hq_crc_ps"][c, s], ("ps state", c, s)


def _fuzz_frame(rng, h, f, wild):
    """frame side info inside what xs_side_info_bad lets through"""
    n_env = int(rng.integers(1, 6))
    if wild == 0:      # a parser's grid: 0 = b0 < .. < b_n = 16
        inner = sorted(rng.choice(np.arange(1, 16), n_env - 1, replace=False).tolist())
        borders = [0] + inner + [16]
    elif wild == 1:    # variable frames: first border 0..3, last 13..19
        lo, hi = int(rng.integers(0, 4)), int(rng.integers(13, 20))
        inner = sorted(rng.choice(np.arange(lo + 1, hi), n_env - 1, replace=False).tolist())
        borders = [lo] + inner + [hi]
    else:              # anything in 0..19: unsorted, repeated (empty envelopes), envelopes wholly behind slot 32
        borders = rng.integers(0, 20, n_env + 1).tolist()
    f.num_env = n_env
    for e in range(9):
        f.border_vec[e] = int(borders[e]) if e < len(borders) else 0
    for e in range(8):
        f.freq_res[e] = int(rng.integers(0, 2))
    f.num_noise_env = 1 if n_env == 1 else 2
    f.noise_border_vec[0] = f.border_vec[0]
    f.noise_border_vec[1] = f.border_vec[(n_env + 1) // 2] if n_env > 1 else f.border_vec[n_env]
    f.noise_border_vec[2] = f.border_vec[n_env]
    f.transient_env = int(rng.integers(-1, n_env + 1))
    for i in range(h.num_if_bands):
        f.sbr_invf_mode[i] = int(rng.integers(0, 4))
    nsf = h.num_sf_bands[1]
    for i in range(nsf):
        f.add_harmonics[i] = int(rng.integers(0, 5) == 0)
    vals = np.frombuffer(f, dtype=np.uint8)  # noqa: F841  (keeps the ctypes buffer alive for the int16 views below)
    for i in range(len(f.int_env_sf_arr)):
        f.int_env_sf_arr[i] = int((int(rng.integers(0, 512)) << 6) | int(rng.integers(0, 40)))
    for i in range(len(f.int_noise_floor)):
        f.int_noise_floor[i] = int((int(rng.integers(0, 512)) << 6) | int(rng.integers(20, 50)))


def test_fuzzed_chains(oracle):
    recs = cap.read_records(GOLDEN)
    rng = np.random.default_rng(4041)
    refused = taken = 0
    for i, r in enumerate(recs):
        st, ps = cap.State.from_buffer_copy(bytes(r["st0"])), cap.PsState.from_buffer_copy(bytes(r["ps0"]))
        h = cap.Header.from_buffer_copy(bytes(r["header"]))
        for step in range(10):
            f = cap.Frame.from_buffer_copy(bytes(r["frame"]))
            _fuzz_frame(rng, h, f, (i + step) % 3)
            if step % 4 == 3:      # the band limit moves: xs_rescale_x_overlap's branches, a change of the adjusted range
                f.max_qmf_subband_aac = int(np.clip(f.max_qmf_subband_aac + rng.integers(-6, 7), h.sub_band_start, 32))
            if step == 6:
                h.smoothing_mode = 1 - h.smoothing_mode
            if step == 8:
                h.limiter_gains = int(rng.integers(0, 4))
            amp = [30000, 3000, 200, 12, 0][step % 5]
            pcm = rng.integers(-amp, amp + 1, 1024).astype(np.int16)
            a, b = both(oracle, h, f, st, r["ps_frame"], ps, pcm)
            same(a, b, (i, step))
            refused += a[0] != 0
            taken += a[0] == 0
            if a[0] == 0:
                st, ps = a[2], a[3]
    assert taken > 150 and refused < taken


# ---- the low-power chain (HE-AACv1: alias reduction, real matrix) -------------------------------------------------------
GOLDEN_LP = os.path.join(ROOT, "tests", "golden", "sbr_lp_records.bin.gz")


def both_lp(oracle, h, f, st, pcm):
    outs = []
    for name in ("xo_sbr_dec_lp", "xo_sbr_dec_lp_seq"):
        s = cap.State.from_buffer_copy(bytes(st))
        out = np.zeros(2048, np.int16)
        rc = getattr(oracle.lib, name)(ctypes.byref(h), ctypes.byref(f), ctypes.byref(s), pcm.ctypes.data_as(P16), 1,
                                       out.ctypes.data_as(P16), 1)
        outs.append((rc, out, s))
    return outs


def same_lp(a, b, tag):
    assert a[0] == b[0], (tag, a[0], b[0])
    if a[0] != 0:
        return
    assert not cap.diff_state(a[2], b[2]), (tag, cap.diff_state(a[2], b[2])[:4])
    assert np.array_equal(a[1], b[1]), (tag, "pcm", int(np.sum(a[1] != b[1])))


def test_lp_reference_records(oracle):
    """low-power frames: paired passes == one-envelope chain == the reference's own outputs"""
    n = 0
    for i, r in enumerate(cap.read_records(GOLDEN_LP)):
        a, b = both_lp(oracle, r["header"], r["frame"], r["st0"], np.ascontiguousarray(r["pcm_in"]))
        same_lp(a, b, i)
        assert a[0] == 0
        assert np.array_equal(a[1], np.asarray(r["pcm_out"]).reshape(-1)[:2048])
        assert not cap.diff_state(a[2], r["st1"])
        n += 1
    assert n >= 24


def test_lp_fuzzed_chains(oracle):
    recs = cap.read_records(GOLDEN_LP)
    rng = np.random.default_rng(5051)
    refused = taken = 0
    for i, r in enumerate(recs):
        st = cap.State.from_buffer_copy(bytes(r["st0"]))
        h = cap.Header.from_buffer_copy(bytes(r["header"]))
        h.interpol_freq = 1 if i % 3 else h.interpol_freq      # most chains take the paired passes
        for step in range(10):
            f = cap.Frame.from_buffer_copy(bytes(r["frame"]))
            _fuzz_frame(rng, h, f, (i + step) % 3)
            if step % 4 == 3:      # the band limit moves: skip != 0 sends the frame through the one-envelope chain
                f.max_qmf_subband_aac = int(np.clip(f.max_qmf_subband_aac + rng.integers(-6, 7), h.sub_band_start, 32))
            if step == 6:
                h.smoothing_mode = 1 - h.smoothing_mode
            if step == 8:
                h.limiter_gains = int(rng.integers(0, 4))
            amp = [30000, 3000, 200, 12, 0][step % 5]
            pcm = rng.integers(-amp, amp + 1, 1024).astype(np.int16)
            a, b = both_lp(oracle, h, f, st, pcm)
            same_lp(a, b, (i, step))
            refused += a[0] != 0
            taken += a[0] == 0
            if a[0] == 0:
                st = a[2]
    assert taken > 150 and refused < taken
