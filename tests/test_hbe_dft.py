"""The DFT harmonic transposer (-esbr_hq:1): ixheaacd_dft_hbe_apply (decoder/ixheaacd_hbe_dft_trans.c:771-941).
CPU: the oracle (oracle/oracle_hbe.cpp: xo_hbe_dft_apply, the sequential run of libxaac_amd/csrc/hbe_dft.h) against the
compiled reference's function on transposers the reference sets up itself from frequency tables, chains of frames.
Tolerance, not bit equality: the transforms are the library's own and the polar part calls libm (include/xaac_hbe.h);
every output row and every carried signal must agree to 2e-5 of the array's peak (float rounding through two transforms
is ~1e-6)."""
import ctypes

import numpy as np
import pytest

from hbe_structs import HbeDftCfg, HbeDftFullState

P16, PF = ctypes.POINTER(ctypes.c_int16), ctypes.POINTER(ctypes.c_float)
REL = 2e-5


def _p(a):
    return a.ctypes.data_as(PF)


def tables(sb, end):
    hi = list(range(sb, end, 2)) + [end]
    lo = hi[::2] if (len(hi) - 1) % 2 == 0 else [hi[0]] + hi[1::2]
    return np.array(lo, np.int16), np.array(hi, np.int16)


# (start band, end band, oversampling, pitch_in_bins): synth_size 12 / 16 / 8, analy_size 28 / 32; the reference has transforms
# for 8 only with oversampling
CASES = [(12, 36, 0, 0), (12, 40, 0, 0), (14, 41, 1, 0), (20, 44, 0, 0), (20, 48, 1, 0), (8, 32, 1, 0), (13, 39, 0, 37),
         (12, 40, 1, 90), (21, 47, 0, 128), (15, 38, 0, 250), (20, 64, 0, 0), (17, 64, 0, 61)]   # the last two: analy_size 48


def close(a, b, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    peak = max(np.abs(a).max(), 1e-30)
    err = np.abs(a - b).max()
    assert np.isfinite(b).all() and err <= REL * peak, (what, err, peak, err / peak)
    return err / peak


def fns(oracle, reference):
    rr, ra, oa = reference.lib.ref_hbe_dft_reinit, reference.lib.ref_hbe_dft_apply, oracle.lib.xo_hbe_dft_apply
    S, C = ctypes.POINTER(HbeDftFullState), ctypes.POINTER(HbeDftCfg)
    rr.restype = ra.restype = oa.restype = ctypes.c_int
    rr.argtypes = [P16, ctypes.c_int, P16, ctypes.c_int, S, C, PF, PF]
    ra.argtypes = [P16, ctypes.c_int, P16, ctypes.c_int, ctypes.c_int, S, PF, PF, ctypes.c_int, ctypes.c_int, PF, PF]
    oa.argtypes = [S, C, PF, PF, PF, PF, ctypes.c_int, ctypes.c_int, PF, PF]
    return rr, ra, oa


def setup(rr, sb, end):
    lo, hi = tables(sb, end)
    st, cfg = HbeDftFullState(), HbeDftCfg()
    coef = [np.zeros((64, 128), np.float32) for _ in range(2)]
    assert rr(lo.ctypes.data_as(P16), len(lo) - 1, hi.ctypes.data_as(P16), len(hi) - 1, ctypes.byref(st), ctypes.byref(cfg), _p(coef[0]),
              _p(coef[1])) == 0
    return lo, hi, st, cfg, coef


def frame_rows(rng, frame, sb):
    """core-band QMF rows: tones + noise below the start band, nothing above (what the transposer is handed)"""
    q = [np.zeros((32, 64), np.float32) for _ in range(2)]
    amp = 2.0 ** rng.integers(2, 13)
    t = np.arange(32)[:, None] + 32 * frame
    for k in rng.integers(1, sb, 3):
        ph = 2 * np.pi * (0.07 * k * t + rng.random())
        q[0][:, k] += (amp * np.cos(ph[:, 0])).astype(np.float32)
        q[1][:, k] += (amp * np.sin(ph[:, 0])).astype(np.float32)
    for a in q:
        a[:, :sb] += (0.05 * amp * rng.standard_normal((32, sb))).astype(np.float32)
    return q


def clone(st):
    c = HbeDftFullState()
    ctypes.memmove(ctypes.byref(c), ctypes.byref(st), ctypes.sizeof(st))
    return c


@pytest.mark.parametrize("case", range(len(CASES)))
def test_oracle_follows_the_reference_through_chains_of_frames(oracle, reference, case):
    rr, ra, oa = fns(oracle, reference)
    sb, end, ovs, pitch = CASES[case]
    lo, hi, st0, cfg, coef = setup(rr, sb, end)
    assert st0.synth_size in (8, 12, 16) and st0.anal.analy_size in (28, 32, 48) and 2 <= st0.max_stretch <= 4, (st0.synth_size, st0.anal.analy_size)
    rng = np.random.default_rng(4000 + case)
    sr, so = clone(st0), clone(st0)
    worst = 0.0
    for frame in range(5):
        q = frame_rows(rng, frame, sb) if frame != 3 else [np.zeros((32, 64), np.float32) for _ in range(2)]
        before = [rng.standard_normal((34, 64)).astype(np.float32) for _ in range(2)]
        pr, po = [a.copy() for a in before], [a.copy() for a in before]
        o = ovs if frame != 1 or sb == 8 else 0       # the flag is the frame's: one frame without it
        assert ra(lo.ctypes.data_as(P16), len(lo) - 1, hi.ctypes.data_as(P16), len(hi) - 1, 0, ctypes.byref(sr), _p(q[0]), _p(q[1]), pitch, o,
                  _p(pr[0]), _p(pr[1])) == 0
        assert oa(ctypes.byref(so), ctypes.byref(cfg), _p(coef[0]), _p(coef[1]), _p(q[0]), _p(q[1]), pitch, o, _p(po[0]), _p(po[1])) == 0
        for nm in ("input_buf", "output_buf", "synth_buf"):
            worst = max(worst, close(np.ctypeslib.as_array(getattr(sr, nm)), np.ctypeslib.as_array(getattr(so, nm)), (nm, frame)))
        worst = max(worst, close(np.ctypeslib.as_array(sr.anal.analy_buf), np.ctypeslib.as_array(so.anal.analy_buf), ("analy_buf", frame)))
        for a, b, nm in zip(pr, po, ("pv_re", "pv_im")):
            worst = max(worst, close(a, b, (nm, frame)))
            untouched = (a == before[0 if nm == "pv_re" else 1])
            assert np.array_equal(untouched, b == before[0 if nm == "pv_re" else 1]), ("cells written", nm, frame)
        if frame in (0, 2):
            assert np.abs(pr[0][:32, st0.anal.a_start:st0.anal.a_start + st0.anal.analy_size]).max() > 0, "the transposer produced nothing"
    assert worst < REL


def test_sizes_without_a_transform_are_refused(oracle, reference):
    rr, ra, oa = fns(oracle, reference)
    lo, hi, st, cfg, coef = setup(rr, 8, 32)      # synth_size 8: no 256-point transform without oversampling
    q = [np.zeros((32, 64), np.float32) for _ in range(2)]
    p = [np.zeros((34, 64), np.float32) for _ in range(2)]
    keep = bytes(st)
    assert oa(ctypes.byref(st), ctypes.byref(cfg), _p(coef[0]), _p(coef[1]), _p(q[0]), _p(q[1]), 0, 0, _p(p[0]), _p(p[1])) == -1
    assert bytes(st) == keep
    sr = clone(st)
    assert ra(lo.ctypes.data_as(P16), len(lo) - 1, hi.ctypes.data_as(P16), len(hi) - 1, 0, ctypes.byref(sr), _p(q[0]), _p(q[1]), 0, 0, _p(p[0]),
              _p(p[1])) != 0     # the reference fails the frame too (ixheaacd_hbe_fft_map, hbe_dft_trans.c:508)


# ---- the committed reference chains (tools/make_golden_hbe_dft.py): oracle here, the GPU library on the box ----------------
import os  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hbe_dft_ref.npz")


def golden_case(g, case):
    sz = [int(v) for v in g["sizes_%d" % case]]
    st = HbeDftFullState()
    st.synth_size, st.k_start, st.start_band, st.end_band, st.max_stretch, st.anal.analy_size, st.anal.a_start = sz
    cfg = HbeDftCfg.from_buffer_copy(g["cfg_%d" % case].tobytes())
    L = sz[5]
    coef = [np.zeros((64, 128), np.float32) for _ in range(2)]
    for c, src in zip(coef, g["coef_%d" % case]):
        c[:L, :2 * L] = src
    return st, cfg, coef


def golden_inputs(case, frame, rng):
    sb, end, ovs, pitch = CASES[case]
    q = frame_rows(rng, frame, sb) if frame != 3 else [np.zeros((32, 64), np.float32) for _ in range(2)]
    before = [rng.standard_normal((34, 64)).astype(np.float32) for _ in range(2)]
    return q, before, (ovs if frame != 1 or sb == 8 else 0), pitch


def state_signals(st):
    return np.concatenate([np.ctypeslib.as_array(st.input_buf), np.ctypeslib.as_array(st.output_buf), np.ctypeslib.as_array(st.synth_buf),
                           np.ctypeslib.as_array(st.anal.analy_buf)])


def test_oracle_reproduces_the_committed_reference_chains(oracle):
    g = np.load(GOLDEN)
    oa = oracle.lib.xo_hbe_dft_apply
    oa.restype = ctypes.c_int
    oa.argtypes = [ctypes.POINTER(HbeDftFullState), ctypes.POINTER(HbeDftCfg), PF, PF, PF, PF, ctypes.c_int, ctypes.c_int, PF, PF]
    for case in g["cases"]:
        case = int(case)
        st, cfg, coef = golden_case(g, case)
        L, a0 = st.anal.analy_size, st.anal.a_start
        rng = np.random.default_rng(4000 + case)
        for frame in range(int(g["frames"])):
            q, before, o, pitch = golden_inputs(case, frame, rng)
            po = [a.copy() for a in before]
            assert oa(ctypes.byref(st), ctypes.byref(cfg), _p(coef[0]), _p(coef[1]), _p(q[0]), _p(q[1]), pitch, o, _p(po[0]), _p(po[1])) == 0
            for k in range(2):
                close(g["rows_%d" % case][frame, k], po[k][:, a0:a0 + L], (case, frame, k))
        close(g["state_%d" % case], state_signals(st), (case, "state"))


@pytest.mark.gpu
def test_gpu_reproduces_the_committed_reference_chains_in_one_batch():
    """every committed chain as one channel of a batch (each with its own configuration), plus copies at other batch positions"""
    import torch
    import libxaac_amd
    g = np.load(GOLDEN)
    cases = [int(c) for c in g["cases"]] * 3
    n = len(cases)
    ctx = libxaac_amd.XaacContext(0)
    sts, cfgs, coefs = zip(*[golden_case(g, c) for c in cases])
    assert ctypes.sizeof(HbeDftFullState) == libxaac_amd.HBE_DFT_FULL_STATE_BYTES and ctypes.sizeof(HbeDftCfg) == libxaac_amd.HBE_DFT_CFG_BYTES
    dev = "cuda:0"
    state = torch.tensor(np.stack([np.frombuffer(bytes(s), np.uint8) for s in sts]), device=dev)
    cfg_tab = torch.tensor(np.stack([np.frombuffer(bytes(c), np.uint8) for c in cfgs]), device=dev)
    coef_re = torch.tensor(np.stack([c[0] for c in coefs]), device=dev)
    coef_im = torch.tensor(np.stack([c[1] for c in coefs]), device=dev)
    cfg_idx = torch.arange(n, dtype=torch.int32, device=dev)
    rngs = [np.random.default_rng(4000 + c) for c in cases]
    status = torch.full((n,), 7, dtype=torch.int32, device=dev)
    for frame in range(int(g["frames"])):
        ins = [golden_inputs(c, frame, r) for c, r in zip(cases, rngs)]
        qre = torch.tensor(np.stack([i[0][0] for i in ins]), device=dev)
        qim = torch.tensor(np.stack([i[0][1] for i in ins]), device=dev)
        pv_re = torch.tensor(np.stack([i[1][0] for i in ins]), device=dev)
        pv_im = torch.tensor(np.stack([i[1][1] for i in ins]), device=dev)
        ovs = torch.tensor([i[2] for i in ins], dtype=torch.int32, device=dev)
        pitch = torch.tensor([i[3] for i in ins], dtype=torch.int32, device=dev)
        ctx.hbe_dft_apply_batch(qre, qim, cfg_tab, coef_re, coef_im, state, pv_re, pv_im, status, pitch_in_bins=pitch, oversampling=ovs, cfg=cfg_idx)
        torch.cuda.synchronize()
        assert (status == 0).all()
        pr, pi = pv_re.cpu().numpy(), pv_im.cpu().numpy()
        for i, c in enumerate(cases):
            L, a0 = sts[i].anal.analy_size, sts[i].anal.a_start
            close(g["rows_%d" % c][frame, 0], pr[i][:, a0:a0 + L], (c, frame, "re", i))
            close(g["rows_%d" % c][frame, 1], pi[i][:, a0:a0 + L], (c, frame, "im", i))
            keep = np.ones((34, 64), bool)
            keep[:32, a0:] = False
            assert np.array_equal(pr[i][keep], ins[i][1][0][keep]), "real rows: only sub-bands a_start.. of rows 0..31 are written"
    host = state.cpu().numpy()
    for i, c in enumerate(cases):
        got = HbeDftFullState.from_buffer_copy(host[i].tobytes())
        close(g["state_%d" % c], state_signals(got), (c, "state", i))


@pytest.mark.gpu
def test_gpu_against_the_oracle_on_fresh_chains_and_refusals(oracle):
    """other tables, levels and pitches than the committed chains: the GPU batch against the oracle channel by channel
    (configurations taken from the committed cases' windows would not fit other tables, so the tables are the committed ones
    and the signals are new); one channel of the batch has sizes without a transform and must be left alone"""
    import torch
    import libxaac_amd
    g = np.load(GOLDEN)
    oa = oracle.lib.xo_hbe_dft_apply
    oa.restype = ctypes.c_int
    oa.argtypes = [ctypes.POINTER(HbeDftFullState), ctypes.POINTER(HbeDftCfg), PF, PF, PF, PF, ctypes.c_int, ctypes.c_int, PF, PF]
    cases = [int(c) for c in g["cases"]]
    n = 4 * len(cases) + 1
    ctx = libxaac_amd.XaacContext(0)
    dev = "cuda:0"
    base = [golden_case(g, c) for c in cases]
    sts = [clone(base[i % len(cases)][0]) for i in range(n)]
    bad = n - 1
    sts[bad].synth_size = 20                      # no transform of 640 points
    cfg_tab = torch.tensor(np.stack([np.frombuffer(bytes(b[1]), np.uint8) for b in base]), device=dev)
    coef_re = torch.tensor(np.stack([b[2][0] for b in base]), device=dev)
    coef_im = torch.tensor(np.stack([b[2][1] for b in base]), device=dev)
    cfg_idx = torch.tensor([i % len(cases) for i in range(n)], dtype=torch.int32, device=dev)
    state = torch.tensor(np.stack([np.frombuffer(bytes(s), np.uint8) for s in sts]), device=dev)
    bad_before = bytes(sts[bad])
    rng = np.random.default_rng(77)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    worst = 0.0
    for frame in range(4):
        qs = [frame_rows(rng, frame, CASES[cases[i % len(cases)]][0]) for i in range(n)]
        befores = [[rng.standard_normal((34, 64)).astype(np.float32) for _ in range(2)] for _ in range(n)]
        ovs = [(0 if sts[i].anal.analy_size == 48 else int(rng.integers(0, 2))) if sts[i].synth_size != 8 else 1 for i in range(n)]  # (the sizes' valid flags)
        pitch = [int(rng.choice([0, 0, 24, 61, 140, 300])) for _ in range(n)]
        qre = torch.tensor(np.stack([q[0] for q in qs]), device=dev)
        qim = torch.tensor(np.stack([q[1] for q in qs]), device=dev)
        pv_re = torch.tensor(np.stack([b[0] for b in befores]), device=dev)
        pv_im = torch.tensor(np.stack([b[1] for b in befores]), device=dev)
        ctx.hbe_dft_apply_batch(qre, qim, cfg_tab, coef_re, coef_im, state, pv_re, pv_im, status,
                                pitch_in_bins=torch.tensor(pitch, dtype=torch.int32, device=dev),
                                oversampling=torch.tensor(ovs, dtype=torch.int32, device=dev), cfg=cfg_idx)
        torch.cuda.synchronize()
        stat = status.cpu().numpy()
        assert stat[bad] == -1 and (stat[:bad] == 0).all()
        pr, pi = pv_re.cpu().numpy(), pv_im.cpu().numpy()
        assert np.array_equal(pr[bad], befores[bad][0]) and np.array_equal(pi[bad], befores[bad][1])
        for i in range(n - 1):
            _, cfg, coef = base[i % len(cases)]
            po = [a.copy() for a in befores[i]]
            assert oa(ctypes.byref(sts[i]), ctypes.byref(cfg), _p(coef[0]), _p(coef[1]), _p(qs[i][0]), _p(qs[i][1]), pitch[i], ovs[i], _p(po[0]),
                      _p(po[1])) == 0
            worst = max(worst, close(po[0], pr[i], (i, frame, "re")), close(po[1], pi[i], (i, frame, "im")))
    host = state.cpu().numpy()
    assert host[bad].tobytes()[:-4] == bad_before[:-4] and HbeDftFullState.from_buffer_copy(host[bad].tobytes()).last_status == -1
    for i in range(n - 1):
        close(state_signals(sts[i]), state_signals(HbeDftFullState.from_buffer_copy(host[i].tobytes())), (i, "state"))
    assert worst < REL


# ---- the host's re-initialisation (libxaac_host.so: xaac_hbe_dft_state_reinit) against the reference's ------------------------
def test_host_reinit_makes_what_the_reference_makes(reference):
    """ixheaacd_dft_hbe_data_reinit (hbe_dft_trans.c:272-455) for every start / end band pair whose sizes the reference has
    transforms for (and a spread of those it has none for): sizes, cross-over bands, the two time windows, the patches'
    cross-over windows and the analysis bank's matrices, word for word"""
    sys_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import sys
    sys.path.insert(0, sys_path)
    from libxaac_amd import decoder
    import sbr_capture as cap
    host = decoder.load_host_library()
    fn = host.xaac_hbe_dft_state_reinit
    fn.restype = ctypes.c_int32
    fn.argtypes = [ctypes.POINTER(HbeDftFullState), ctypes.POINTER(HbeDftCfg), PF, PF, ctypes.POINTER(cap.Header)]
    rr = reference.lib.ref_hbe_dft_reinit
    rr.restype = ctypes.c_int
    rr.argtypes = [P16, ctypes.c_int, P16, ctypes.c_int, ctypes.POINTER(HbeDftFullState), ctypes.POINTER(HbeDftCfg), PF, PF]
    same = refused = 0
    for sb in range(5, 28):
        for end in range(sb + 12, min(sb + 36, 65)):
            lo, hi = tables(sb, end)
            if len(lo) - 1 > 28 or len(hi) - 1 > 56:
                continue
            for prev_ms in (0, 3):
                sr, so, cr, co = HbeDftFullState(), HbeDftFullState(), HbeDftCfg(), HbeDftCfg()
                sr.max_stretch = so.max_stretch = prev_ms
                kr = [np.zeros((64, 128), np.float32) for _ in range(2)]
                ko = [np.full((64, 128), 7, np.float32) for _ in range(2)]
                rc_r = rr(lo.ctypes.data_as(P16), len(lo) - 1, hi.ctypes.data_as(P16), len(hi) - 1, ctypes.byref(sr), ctypes.byref(cr), _p(kr[0]), _p(kr[1]))
                h = cap.Header()
                h.num_sf_bands[0], h.num_sf_bands[1] = len(lo) - 1, len(hi) - 1
                for i, v in enumerate(lo):
                    h.freq_band_tbl_lo[i] = int(v)
                for i, v in enumerate(hi):
                    h.freq_band_tbl_hi[i] = int(v)
                rc_o = fn(ctypes.byref(so), ctypes.byref(co), _p(ko[0]), _p(ko[1]), ctypes.byref(h))
                fits = rc_r == 0 and 32 * sr.synth_size in (256, 384, 512) and 16 * sr.anal.analy_size in (448, 512, 768)
                if not fits:
                    refused += rc_o != 0
                    continue
                assert rc_o == 0, (sb, end, prev_ms)
                for nm in ("synth_size", "k_start", "start_band", "end_band", "max_stretch"):
                    assert getattr(sr, nm) == getattr(so, nm), (nm, sb, end)
                assert (sr.anal.analy_size, sr.anal.a_start) == (so.anal.analy_size, so.anal.a_start) and list(sr.x_over_qmf) == list(so.x_over_qmf)
                assert bytes(cr) == bytes(co), ("windows", sb, end, prev_ms)
                assert np.array_equal(kr[0].view(np.uint32), ko[0].view(np.uint32)) and np.array_equal(kr[1].view(np.uint32), ko[1].view(np.uint32))
                same += 1
    assert same > 150 and refused > 50, (same, refused)


@pytest.mark.gpu
def test_path_a_chain_with_the_dft_transposer_is_the_same_for_a_batch_and_for_its_channels_alone(oracle):
    """xaac_esbr_sbr_process_batch with hbe_dft_state on the committed reference-made Path A chains (HE-AAC channels with their
    harmonic frames): every chain gets the DFT transposer the host's re-initialisation makes from its header's band tables (a
    new one where the tables change), chains whose sizes have no transform stay in the batch as refused channels.  The whole batch
    walked step by step against every chain walked alone: outputs, Path A states and transposer states byte for byte -- the
    per-channel strides, configuration indices, skipped (no SBR processing) and refused channels of the chained launches.
    And every chain walked by the oracle (oracle/oracle_esbr.cpp: xo_esbr_sbr_frame_dft, the sequential run of the same headers):
    the frames' 2048 output samples to 1e-4 of their peak, the return codes equal.
    (What the chain computes is held to the reference through the drop-in: tests/test_dropin_gpu.py, tests/test_sweep_gpu.py.)"""
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tools"))
    import libxaac_amd
    from libxaac_amd import decoder
    import sbr_capture as cap
    import test_esbr_chains as tc
    from make_golden_esbr_chains import chain_core
    host = decoder.load_host_library()
    reinit = host.xaac_hbe_dft_state_reinit
    reinit.restype = ctypes.c_int32
    reinit.argtypes = [ctypes.POINTER(HbeDftFullState), ctypes.POINTER(HbeDftCfg), PF, PF, ctypes.c_void_p]
    CH = tc.FIXTURES["aac"]
    order = tc.steps_of_chains(CH)
    chains = [c for c in range(len(order)) if not CH["chain_ps"][c] and tc.chain_has_transposer(CH, order[c]) and len(order[c]) >= 8][:20]
    n = len(chains)
    assert n >= 8
    spoiled = {chains[3]}       # one channel whose transposer gets a bank size without a transform: refused inside the batch
    dev = torch.device("cuda:0")
    ctx = libxaac_amd.XaacContext(0, 0)
    steps = 8

    def tables_of(row):
        h = cap.Header.from_buffer_copy(CH["header"][row].tobytes())
        return bytes(h)[cap.Header.num_sf_bands.offset:cap.Header.freq_band_tbl_noise.offset]

    def walk(members):
        """the chains `members` as one batch; returns per step (out, state, dft state) arrays"""
        m = len(members)
        dst = [HbeDftFullState() for _ in members]
        cfgs = [HbeDftCfg() for _ in members]
        coefs = [[np.zeros((64, 128), np.float32) for _ in range(2)] for _ in members]
        seen = [None] * m
        t_st = torch.from_numpy(np.ascontiguousarray(CH["est0"][members])).to(dev)
        t_d = None
        trace = []
        for s in range(steps):
            rows = [order[c][s] for c in members]
            host_d = t_d.cpu().numpy() if t_d is not None else None
            for i, r in enumerate(rows):                      # a header with other band tables: a new transposer, signals kept
                tb = tables_of(r)
                if tb != seen[i]:
                    seen[i] = tb
                    if host_d is not None:
                        dst[i] = HbeDftFullState.from_buffer_copy(host_d[i].tobytes())
                    hdr = np.ascontiguousarray(CH["header"][r])
                    reinit(ctypes.byref(dst[i]), ctypes.byref(cfgs[i]), _p(coefs[i][0]), _p(coefs[i][1]), hdr.ctypes.data_as(ctypes.c_void_p))
                    if members[i] in spoiled:
                        dst[i].synth_size = 20
                    if host_d is not None:
                        host_d[i] = np.frombuffer(bytes(dst[i]), np.uint8)
            if t_d is None:
                host_d = np.stack([np.frombuffer(bytes(d), np.uint8) for d in dst])
            t_d = torch.from_numpy(np.ascontiguousarray(host_d)).to(dev)
            cfg_tab = torch.from_numpy(np.stack([np.frombuffer(bytes(c), np.uint8) for c in cfgs])).to(dev)
            cre = torch.from_numpy(np.stack([k[0] for k in coefs])).to(dev)
            cim = torch.from_numpy(np.stack([k[1] for k in coefs])).to(dev)
            core = torch.from_numpy(np.stack([chain_core(int(CH["chain_run"][c]), int(CH["chain_id"][c]), s) for c in members])).to(dev)
            g = lambda k: torch.from_numpy(np.ascontiguousarray(CH[k][rows])).to(dev)
            out = torch.zeros((m, 2048), dtype=torch.float32, device=dev)
            status = torch.full((m,), 7, dtype=torch.int32, device=dev)
            ws = torch.zeros(ctx.esbr_workspace_bytes(m), dtype=torch.uint8, device=dev)
            ctx.esbr_sbr_process_batch(core, g("header"), g("frame"), g("side"), t_st, out, ws, status,
                                       hbe_dft=(t_d, cfg_tab, cre, cim, torch.arange(m, dtype=torch.int32, device=dev)))
            ctx.sync()
            trace.append((out.cpu().numpy(), t_st.cpu().numpy(), t_d.cpu().numpy(), status.cpu().numpy()))
        return trace

    whole = walk(chains)
    ofn = oracle.lib.xo_esbr_sbr_frame_dft
    ofn.restype = ctypes.c_int
    ofn.argtypes = [PF] + [ctypes.c_void_p] * 4 + [PF, ctypes.POINTER(HbeDftFullState), ctypes.POINTER(HbeDftCfg), PF, PF]

    def oracle_walk(c):
        dst, cfg = HbeDftFullState(), HbeDftCfg()
        coef = [np.zeros((64, 128), np.float32) for _ in range(2)]
        st = np.ascontiguousarray(CH["est0"][c]).copy()
        seen, outs = None, []
        for s in range(steps):
            r = order[c][s]
            if tables_of(r) != seen:
                seen = tables_of(r)
                hdr = np.ascontiguousarray(CH["header"][r])
                reinit(ctypes.byref(dst), ctypes.byref(cfg), _p(coef[0]), _p(coef[1]), hdr.ctypes.data_as(ctypes.c_void_p))
                if c in spoiled:
                    dst.synth_size = 20
            core = np.ascontiguousarray(chain_core(int(CH["chain_run"][c]), int(CH["chain_id"][c]), s))
            out = np.zeros(2048, np.float32)
            vp = lambda a: np.ascontiguousarray(a).ctypes.data_as(ctypes.c_void_p)
            hd, fr, sd = (np.ascontiguousarray(CH[k][r]) for k in ("header", "frame", "side"))
            rc = ofn(_p(core), vp(hd), vp(fr), vp(sd), st.ctypes.data_as(ctypes.c_void_p), _p(out), ctypes.byref(dst), ctypes.byref(cfg), _p(coef[0]),
                     _p(coef[1]))
            outs.append((out, rc, dst.last_status))
        return outs
    ran = refused = 0
    worst = 0.0
    for i, c in enumerate(chains):
        alone = walk([c])
        orc = oracle_walk(c)
        for s in range(steps):
            o_out, o_rc, o_last = orc[s]
            assert int(whole[s][3][i]) == o_rc and HbeDftFullState.from_buffer_copy(whole[s][2][i].tobytes()).last_status == o_last, (c, s, o_rc, o_last)
            peak = max(np.abs(o_out).max(), 1e-20)
            err = np.abs(whole[s][0][i] - o_out).max() / peak
            worst = max(worst, err)
            assert err <= 1e-4, ("oracle", c, s, err, peak)
            for k, nm in enumerate(("out", "state", "transposer state", "status")):
                assert np.array_equal(whole[s][k][i], alone[s][k][0]), (nm, c, s)
            last = HbeDftFullState.from_buffer_copy(whole[s][2][i].tobytes()).last_status
            ran += last == 0
            refused += last == -1
    assert ran > 100 and refused == steps, (ran, refused)
    # the transposer's rows reach the output where a frame asks for harmonic patching: such frames exist in the chains walked
    from esbr_structs import EsbrSide
    off = EsbrSide.harmonic_sbr.offset
    assert any(int(np.frombuffer(CH["side"][order[c][s]].tobytes()[off:off + 2], np.int16)[0]) & 1 for c in chains for s in range(steps))
    ctx.close()
