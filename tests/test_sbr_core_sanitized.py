"""The shared SBR core (libxaac_amd/csrc/sbr_core.h, sbr_ps*.h: what the GPU kernels and the oracle both compile) under
AddressSanitizer on the host, its QMF matrices on the stack (oracle_sbr.cpp: XO_MATRIX_ON_STACK): low-power and HQ + PS chains
over frames with envelope grids of every kind -- unsorted and empty envelopes, grids that end before slot 16 or start behind
slot 32 -- and a band limit that moves.  Side info no parser produces, but the boundary does not trust its caller: a row
index in front of the matrix is the neighbouring wave's LDS on the GPU (round 5 found xs_rescale_x_overlap clearing such
rows).  CPU only."""
import glob
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


def test_fuzzed_grids_and_moving_band_limit_touch_nothing_outside_the_matrices(tmp_path):
    lib = str(tmp_path / "oracle_sbr_asan.so")
    srcs = [os.path.join(ROOT, "oracle", "oracle_sbr.cpp")] + sorted(glob.glob(os.path.join(ROOT, "oracle", "oracle_qmf*.cpp")))
    csrc = os.path.join(ROOT, "oracle", "oracle_imdct.c")
    obj = str(tmp_path / "oracle_c.o")
    subprocess.check_call(["gcc", "-O1", "-std=c99", "-fPIC", "-c", csrc, "-o", obj])
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-w", "-DXO_MATRIX_ON_STACK",
                           "-fsanitize=address", "-fno-omit-frame-pointer", *srcs, obj, "-o", lib])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
    code = CHILD % {"tests": os.path.join(ROOT, "tests"), "lib": lib, "golden": os.path.join(ROOT, "tests", "golden")}
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and "AddressSanitizer" not in p.stderr, (p.stdout[-300:], p.stderr[-3000:])
    assert "calls 1440" in p.stdout
