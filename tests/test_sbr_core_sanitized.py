"""The shared SBR core (libxaac_amd/csrc/sbr_core.h, sbr_ps*.h: what the GPU kernels and the oracle both compile) under
AddressSanitizer on the host, its QMF matrices on the stack (oracle_sbr.cpp: XO_MATRIX_ON_STACK): low-power and HQ + PS chains
over frames with envelope grids of every kind -- unsorted and empty envelopes, grids that end before slot 16 or start behind
slot 32 -- and a band limit that moves.  Side info no parser produces, but the boundary does not trust its caller: a row
index in front of the matrix is the neighbouring wave's LDS on the GPU (round 5 found xs_rescale_x_overlap clearing such
rows).  CPU only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, sys
import numpy as np
sys.path.insert(0, %(tests)r)
import sbr_capture as cap
from test_env_pairs_cpu import _fuzz_frame
lib = ctypes.CDLL(%(lib)r)
P16 = ctypes.POINTER(ctypes.c_int16)
calls = 0
for mode in ("lp", "hq"):
    recs = cap.read_records(%(golden)r + ("/sbr_lp_records.bin.gz" if mode == "lp" else "/sbr_hq_ps_records.bin.gz"))
    rng = np.random.default_rng(5)
    states = [cap.State.from_buffer_copy(bytes(r["st0"])) for r in recs]
    ps = [cap.PsState.from_buffer_copy(bytes(r["ps0"])) for r in recs] if mode == "hq" else None
    hdrs = [cap.Header.from_buffer_copy(bytes(r["header"])) for r in recs]
    for step in range(12):
        for i, r in enumerate(recs):
            f = cap.Frame.from_buffer_copy(bytes(r["frame"]))
            _fuzz_frame(rng, hdrs[i], f, (i + step) %% 3)
            if step %% 2 == 1:
                f.max_qmf_subband_aac = int(np.clip(f.max_qmf_subband_aac + rng.integers(-6, 7), hdrs[i].sub_band_start, 32))
            pcm = rng.integers(-3000, 3000, 1024).astype(np.int16)
            out = np.zeros(4096, np.int16)
            if mode == "lp":
                lib.xo_sbr_dec_lp(ctypes.byref(hdrs[i]), ctypes.byref(f), ctypes.byref(states[i]), pcm.ctypes.data_as(P16), 1,
                                  out.ctypes.data_as(P16), 1)
            else:
                lib.xo_sbr_dec_hq(ctypes.byref(hdrs[i]), ctypes.byref(f), ctypes.byref(states[i]), ctypes.byref(r["ps_frame"]),
                                  ctypes.byref(ps[i]), pcm.ctypes.data_as(P16), 1, out.ctypes.data_as(P16), 2)
            calls += 1
print("calls", calls)
'''


@pytest.fixture(scope="module")
def asan_oracle(tmp_path_factory):
    """the oracle's SBR / eSBR / transposer / PVC sources with their matrices on the stack, under AddressSanitizer"""
    d = tmp_path_factory.mktemp("asan")
    lib, obj = str(d / "oracle_sbr_asan.so"), str(d / "oracle_c.o")
    srcs = [os.path.join(ROOT, "oracle", "oracle_%s.cpp" % n) for n in ("sbr", "qmf", "esbr", "hbe", "pvc")]
    subprocess.check_call(["gcc", "-O1", "-std=c99", "-fPIC", "-c", os.path.join(ROOT, "oracle", "oracle_imdct.c"), "-o", obj])
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-w", "-DXO_MATRIX_ON_STACK",
                           "-fsanitize=address", "-fno-omit-frame-pointer", *srcs, obj, "-o", lib])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return lib, dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")


def run_child(asan_oracle, template):
    lib, env = asan_oracle
    code = template % {"tests": os.path.join(ROOT, "tests"), "tools": os.path.join(ROOT, "tools"), "lib": lib,
                       "golden": os.path.join(ROOT, "tests", "golden")}
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and "AddressSanitizer" not in p.stderr, (p.stdout[-300:], p.stderr[-3000:])
    return p.stdout


def test_fuzzed_grids_and_moving_band_limit_touch_nothing_outside_the_matrices(asan_oracle):
    assert "calls 1440" in run_child(asan_oracle, CHILD)


CHILD_ESBR = r'''
import ctypes, sys
import numpy as np
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(tools)r)
import sbr_capture as cap
from test_env_pairs_cpu import _fuzz_frame
from make_golden_esbr_chains import chain_core
PF = ctypes.POINTER(ctypes.c_float)
CH = np.load(%(golden)r + "/esbr_chains.npz")
lib = ctypes.CDLL(%(lib)r)
fn = lib.xo_esbr_sbr_frame_hbe
fn.restype = ctypes.c_int
fn.argtypes = [PF] + [ctypes.c_void_p] * 6 + [PF, PF, ctypes.c_void_p]
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
sc = CH["step_chain"]
rng = np.random.default_rng(3)
calls = refused = 0
for c in range(len(CH["chain_len"])):
    rows = np.nonzero(sc == c)[0]
    run, cid, eps = int(CH["chain_run"][c]), int(CH["chain_id"][c]), bool(CH["chain_ps"][c])
    st, hb, ps = CH["est0"][c].copy(), CH["hbs0"][c].copy(), CH["eps0"][c].copy()
    for s, r in enumerate(rows[:6]):
        core = np.ascontiguousarray(chain_core(run, cid, s))
        out, out_r = np.zeros(2048, np.float32), np.zeros(2048, np.float32)
        h, f, sd, pf = (np.ascontiguousarray(CH[k][r]).copy() for k in ("header", "frame", "side", "ps_frame"))
        hh, ff = cap.Header.from_buffer(h), cap.Frame.from_buffer(f)
        if s %% 2 == 1:
            _fuzz_frame(rng, hh, ff, (c + s) %% 3)
        if s %% 4 == 3:
            ff.max_qmf_subband_aac = int(np.clip(ff.max_qmf_subband_aac + rng.integers(-6, 7), hh.sub_band_start, 32))
        rc = fn(core.ctypes.data_as(PF), vp(h), vp(f), vp(sd), vp(st), vp(pf) if eps else None, vp(ps) if eps else None,
                out.ctypes.data_as(PF), out_r.ctypes.data_as(PF) if eps else None, vp(hb))
        calls += 1
        refused += rc != 0
print("calls", calls, "refused", refused)
'''


def test_path_a_chain_on_fuzzed_grids_touches_nothing_outside_its_matrices(asan_oracle):
    """the float chain (eSBR HF generator, envelope adjuster, float PS, transposer in the chain: esbr_core.h, esbr_ps.h,
    hbe_*.h) on the reference-made chains' frames with every second frame's grid fuzzed and the band limit moving"""
    out = run_child(asan_oracle, CHILD_ESBR)
    n, r = (int(t) for t in out.split()[1::2])
    assert n > 250 and r < n // 2


CHILD_RATIO = r'''
import ctypes, sys
import numpy as np
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(tools)r)
import sbr_capture as cap
from test_env_pairs_cpu import _fuzz_frame
from make_golden_esbr_chains import chain_core
PF = ctypes.POINTER(ctypes.c_float)
CH = np.load(%(golden)r + "/esbr_ratio_chains.npz")
lib = ctypes.CDLL(%(lib)r)
fn = lib.xo_esbr_sbr_frame_ratio
fn.restype = ctypes.c_int
fn.argtypes = [PF, ctypes.c_int] + [ctypes.c_void_p] * 6 + [PF, PF, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
sc = CH["step_chain"]
rng = np.random.default_rng(3)
calls = refused = 0
for c in range(len(CH["chain_len"])):
    rows = np.nonzero(sc == c)[0]
    run, cid, ratio = int(CH["chain_run"][c]), int(CH["chain_id"][c]), int(CH["chain_ratio"][c])
    st, pv = CH["est0"][c].copy(), CH["pvst0"][c].copy()
    for s, r in enumerate(rows[:6]):
        core = np.ascontiguousarray(chain_core(run, cid, s))
        out = np.zeros(4096, np.float32)
        h, f, sd, ps = (np.ascontiguousarray(CH[k][r]).copy() for k in ("header", "frame", "side", "pvc_side"))
        hh, ff = cap.Header.from_buffer(h), cap.Frame.from_buffer(f)
        if s %% 2 == 1:
            _fuzz_frame(rng, hh, ff, (c + s) %% 3)
        if s %% 4 == 3:
            ff.max_qmf_subband_aac = int(np.clip(ff.max_qmf_subband_aac + rng.integers(-6, 7), hh.sub_band_start, 32))
        rc = fn(core.ctypes.data_as(PF), ratio, vp(h), vp(f), vp(sd), vp(st), None, None, out.ctypes.data_as(PF), None, None, vp(ps), vp(pv))
        calls += 1
        refused += rc != 0
print("calls", calls, "refused", refused)
'''


def test_8_3_and_4_1_chains_on_fuzzed_grids_touch_nothing_outside_their_matrices(asan_oracle):
    """the same at the other two SBR ratios (tests/golden/esbr_ratio_chains.npz): 4:1's rows are four to a border and its matrices
    the 64-slot ones; PVC frames with the PVC decoder in the chain"""
    out = run_child(asan_oracle, CHILD_RATIO)
    n, r = (int(t) for t in out.split()[1::2])
    assert n > 250 and r < n // 2


CHILD_DFT = r'''
import ctypes, sys
import numpy as np
sys.path.insert(0, %(tests)r)
import test_hbe_dft as t
from hbe_structs import HbeDftCfg, HbeDftFullState
PF = ctypes.POINTER(ctypes.c_float)
G = np.load(t.GOLDEN)
lib = ctypes.CDLL(%(lib)r)
fn = lib.xo_hbe_dft_apply
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.POINTER(HbeDftFullState), ctypes.POINTER(HbeDftCfg), PF, PF, PF, PF, ctypes.c_int, ctypes.c_int, PF, PF]
rng = np.random.default_rng(11)
calls = refused = 0
for case in [int(c) for c in G["cases"]]:
    st0, cfg, coef = t.golden_case(G, case)
    for trial in range(14):
        st = t.clone(st0)
        if trial %% 3 == 1:     # sizes and offsets the tables would never give: refused, or run inside the buffers
            st.k_start = int(rng.integers(-2, 40)); st.anal.a_start = int(rng.integers(-2, 70)); st.max_stretch = int(rng.integers(-1, 7))
        if trial %% 5 == 4:
            st.synth_size = int(rng.choice([4, 8, 12, 16, 20, 7])); st.anal.analy_size = int(rng.choice([4, 24, 28, 32, 36, 64]))
        for frame in range(2):
            q = [np.ascontiguousarray(a) for a in t.frame_rows(rng, frame, t.CASES[case][0])]
            if trial %% 4 == 2:   # values no decoder produces: the index arithmetic must not follow them anywhere
                q[0][rng.integers(0, 32), rng.integers(0, 64)] = np.float32(rng.choice([np.inf, -np.inf, np.nan, 3e38]))
            pv = [np.zeros((34, 64), np.float32) for _ in range(2)]
            pitch = int(rng.choice([0, 1, 11, 12, 60, 127, 4000, -5, 2 ** 20]))
            rc = fn(ctypes.byref(st), ctypes.byref(cfg), t._p(coef[0]), t._p(coef[1]), t._p(q[0]), t._p(q[1]), pitch, int(rng.integers(0, 2)),
                    t._p(pv[0]), t._p(pv[1]))
            calls += 1
            refused += rc != 0
print("calls", calls, "refused", refused)
'''


def test_dft_transposer_on_fuzzed_sizes_pitches_and_non_finite_rows_touches_nothing_outside_its_arrays(asan_oracle):
    """xo_hbe_dft_apply (hbe_dft.h, the code the kernel runs) under AddressSanitizer: the committed configurations with start
    bands, sizes and stretch counts no table gives, pitches far outside the 7 bits the stream carries, infinities and NaNs in
    the rows -- refused or run, never outside a buffer"""
    out = run_child(asan_oracle, CHILD_DFT)
    n, r = (int(t) for t in out.split()[1::2])
    assert n >= 7 * 14 * 2 and n % 28 == 0 and 0 < r < n
