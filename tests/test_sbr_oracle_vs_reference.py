"""Pins the low-power SBR oracle to the compiled reference on (a) every ixheaacd_sbr_dec call captured while the
reference decodes freshly encoded HE-AACv1 streams and (b) the same frames with fuzzed side info pushed through
the reference by oracle/ref_sbr_adapter.c, with the state chained through the reference.  Needs oracle/_ref."""
import ctypes
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

import sbr_capture as cap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P16 = ctypes.POINTER(ctypes.c_int16)


@pytest.fixture(scope="module")
def captures(reference, tmp_path_factory):
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "xaacdec_capture")):
        pytest.skip("capture build of the reference decoder missing")
    d = tmp_path_factory.mktemp("streams")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_test_streams.py"), str(d), "5"],
                          stdout=subprocess.DEVNULL)
    files = sorted(glob.glob(os.path.join(str(d), "*aot5*.cap")))
    assert len(files) >= 6
    return files


def _run(lib, fn, h, f, st, pin):
    out = np.zeros(2048, np.int16)
    rc = getattr(lib, fn)(ctypes.byref(h), ctypes.byref(f), ctypes.byref(st), pin.ctypes.data_as(P16), 1,
                          out.ctypes.data_as(P16), 1)
    return rc, out


def test_captured_calls_bit_exact(oracle, reference, captures):
    n = 0
    for path in captures:
        for r in cap.read_records(path):
            st = cap.State.from_buffer_copy(bytes(r["st0"]))
            rc, out = _run(oracle.lib, "xo_sbr_dec_lp", r["header"], r["frame"], st, np.ascontiguousarray(r["pcm_in"]))
            assert rc == r["ret"] and np.array_equal(out, r["pcm_out"][0]), (path, r["call"])
            assert not cap.diff_state(st, r["st1"]), (path, r["call"], cap.diff_state(st, r["st1"])[:3])
            n += 1
    assert n > 1000


def test_fuzzed_side_info_chained_through_reference(oracle, reference, captures):
    rng = np.random.default_rng(5)
    n = 0
    for path in captures[::2]:
        chains = {}
        for r in cap.read_records(path):
            key = r["call"] & 1
            h = cap.Header.from_buffer_copy(bytes(r["header"]))
            f = cap.Frame.from_buffer_copy(bytes(r["frame"]))
            for i in range(h.num_if_bands):
                f.sbr_invf_mode[i] = int(rng.integers(0, 4))
            h.limiter_gains = int(rng.integers(0, 4))
            h.interpol_freq = int(rng.integers(0, 2))
            h.smoothing_mode = int(rng.integers(0, 2))
            if rng.integers(0, 4) == 0:
                for i in range(h.num_sf_bands[1]):
                    f.add_harmonics[i] = int(rng.integers(0, 3) == 0)
            st_r = chains.get(key) or cap.State.from_buffer_copy(bytes(r["st0"]))
            st_o = cap.State.from_buffer_copy(bytes(st_r))
            pin = np.ascontiguousarray(r["pcm_in"])
            ra, oa = _run(reference.lib, "ref_sbr_dec_lp", h, f, st_r, pin)
            rb, ob = _run(oracle.lib, "xo_sbr_dec_lp", h, f, st_o, pin)
            assert ra == rb and np.array_equal(oa, ob), (path, r["call"])
            assert not cap.diff_state(st_o, st_r), (path, r["call"], cap.diff_state(st_o, st_r)[:3])
            chains[key] = st_r
            n += 1
    assert n > 500
