"""Float parametric stereo of the reference's default SBR path: the oracle (oracle/oracle_esbr.cpp: xo_esbr_apply_ps,
arithmetic of libxaac_amd/csrc/esbr_ps.h, mixing matrices from the libm-made table of tools/gen_tables_esbr_ps.py)
against the compiled reference's own ixheaacd_esbr_apply_ps (decoder/ixheaacd_ps_dec_flt.c:389, driven by
oracle/ref_esbr_adapter.c), on chains of frames with the whole tool state carried: the PS side info of the captured
HE-AACv2 stream (1-4 envelopes) and fuzzed IID / ICC indices, both quantisers, 1-5 envelopes with random borders, every
IPD band limit, moving upper band limits.  Both channels' matrices and every byte of the state identical as raw words."""
import ctypes
import os

import numpy as np
import pytest

import sbr_capture as c
from esbr_structs import EsbrPsState, new_ps_state

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PF = ctypes.POINTER(ctypes.c_float)


def bind(lib, name):
    fn = getattr(lib, name)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, PF, PF, PF, PF, ctypes.c_int]
    return fn


def fuzz_ps_frame(rng, pf, mode):
    """mode 0: the captured frame as it is (num_env recovered from its borders); 1: random indices on the captured grid;
    2: 1..5 envelopes with random borders"""
    if mode == 2:
        ne = int(rng.integers(1, 6))
        cuts = sorted(rng.choice(np.arange(1, 32), ne - 1, replace=False).tolist()) if ne > 1 else []
        b = [0] + cuts + [32]
        for i in range(7):
            pf.border_position[i] = b[i] if i < len(b) else 32
        pf.num_env = ne
    else:
        ne = next(e for e in range(1, 7) if pf.border_position[e] >= 32)
        pf.num_env = ne
    if mode:
        pf.iid_quant = int(rng.integers(0, 2))
        lim = 15 if pf.iid_quant else 7
        for e in range(pf.num_env):
            for b_ in range(20):
                pf.iid_par_table[e][b_] = int(rng.integers(-lim, lim + 1))
                pf.icc_par_table[e][b_] = int(rng.integers(0, 8))
    pf.freq_res_ipd = int(rng.integers(0, 3))
    return pf


def matrices(rng, level, usb):
    l = np.zeros((2, 38, 64), np.float32)
    l[:, :, :usb] = (rng.standard_normal((2, 38, usb)) * level).astype(np.float32)
    l[:, 32:, 5:] = 0                     # the look-ahead rows carry bands 0..4 only (sbr_dec.c:487-505)
    l[0, :, 1] += np.float32(level * 6) * np.cos(np.arange(38) * 0.9).astype(np.float32)
    if rng.integers(0, 4) == 0:
        l[:, 10:14] *= np.float32(40)     # a transient for the detector
    return np.ascontiguousarray(l[0]), np.ascontiguousarray(l[1])


@pytest.mark.parametrize("seed", range(8))
def test_apply_ps_chain(oracle, reference, seed):
    ref_fn, orc_fn = bind(reference.lib, "ref_esbr_apply_ps"), bind(oracle.lib, "xo_esbr_apply_ps")
    recs = [r for r in c.read_records(os.path.join(ROOT, "tests", "golden", "sbr_hq_ps_records.bin.gz")) if r["ps"]]
    rng = np.random.default_rng(500 + seed)
    st_r, st_o = new_ps_state(), new_ps_state()
    start = int(rng.integers(0, len(recs) - 30))
    for n, rec in enumerate(recs[start:start + 30]):
        pf = fuzz_ps_frame(rng, c.PsFrame.from_buffer_copy(bytes(rec["ps_frame"])), seed % 3)
        usb = int(rng.choice([64, 48, 41, 35, 30]))
        lre, lim = matrices(rng, float(2.0 ** rng.integers(0, 14)), usb)
        outs = []
        for fn, st in ((ref_fn, st_r), (orc_fn, st_o)):
            a, b = lre.copy(), lim.copy()
            rre, rim = np.zeros((32, 64), np.float32), np.zeros((32, 64), np.float32)
            fn(ctypes.byref(pf), ctypes.byref(st), a.ctypes.data_as(PF), b.ctypes.data_as(PF), rre.ctypes.data_as(PF),
               rim.ctypes.data_as(PF), usb)
            outs.append((a, b, rre, rim))
        for i, nm in enumerate(("left re", "left im", "right re", "right im")):
            x, y = outs[0][i], outs[1][i]
            bad = np.argwhere(x[:32].view(np.uint32) != y[:32].view(np.uint32))
            assert bad.size == 0, (n, nm, len(bad), bad[:4].tolist(), [(float(x[tuple(q)]), float(y[tuple(q)])) for q in bad[:3]])
        if bytes(st_r) != bytes(st_o):
            raise AssertionError((n, c.diff_state(st_r, st_o)))
        assert np.any(outs[0][2] != 0)
