"""The AAC-LD / ELD IMDCT restatement (libxaac_amd/csrc/imdct_ld.h through oracle/oracle_imdct_ld.cpp) against the compiled
reference's ixheaacd_imdct_process with frame_length 512 / 480 and object types 23 / 39: chains of frames with the overlap
carried, both LD window shapes on either side, levels from silence to full scale."""
import ctypes

import numpy as np
import pytest

P32 = ctypes.POINTER(ctypes.c_int32)
P16 = ctypes.POINTER(ctypes.c_int16)


def bind(reference, oracle):
    rf = reference.lib.ref_imdct_ld_process
    rf.restype = ctypes.c_int
    rf.argtypes = [P32, P32, P16, ctypes.c_int, ctypes.c_int, ctypes.c_int, P16, ctypes.c_int]
    of = oracle.lib.xo_imdct_ld_process
    of.restype = ctypes.c_int
    of.argtypes = [P32, P32, P16, ctypes.c_int, ctypes.c_int, ctypes.c_int, P16, ctypes.c_int]
    return rf, of


def spectrum(rng, n, level, kind):
    if kind == 1:
        x = np.zeros(n, np.int64)
        x[rng.integers(0, n, 6)] = rng.integers(-level, level + 1, 6)
    else:
        x = rng.integers(-level, level + 1, n)
    return x.astype(np.int32)


@pytest.mark.parametrize("frame_length", [512, 480])
@pytest.mark.parametrize("eld", [0, 1])
def test_chains_vs_reference(oracle, reference, frame_length, eld):
    rf, of = bind(reference, oracle)
    rng = np.random.default_rng(frame_length + eld)
    n_ov = 3 * frame_length if eld else frame_length // 2
    for chain in range(6):
        ov_r, ov_o = np.zeros(2048, np.int32), np.zeros(n_ov, np.int32)
        ps_r, ps_o = np.zeros(1, np.int16), np.zeros(1, np.int16)
        for frame in range(24):
            level = [0, 1, 300, 2 ** 17, 2 ** 24, 2 ** 30, 2 ** 31 - 1][(frame + chain) % 7]
            spec = spectrum(rng, frame_length, level, frame % 3)
            if level == 2 ** 31 - 1 and frame % 2:
                spec[:] = -2 ** 31
            shape = int(rng.integers(0, 2))
            sr = np.zeros(2048, np.int32)
            sr[:frame_length] = spec
            pr, po = np.zeros(frame_length, np.int16), np.zeros(frame_length, np.int16)
            qr = rf(sr.ctypes.data_as(P32), ov_r.ctypes.data_as(P32), ps_r.ctypes.data_as(P16), shape, frame_length, 39 if eld else 23,
                    pr.ctypes.data_as(P16), 1)
            so = spec.copy()
            qo = of(so.ctypes.data_as(P32), ov_o.ctypes.data_as(P32), ps_o.ctypes.data_as(P16), shape, frame_length, eld,
                    po.ctypes.data_as(P16), 1)
            assert np.array_equal(so, spec) and qr == qo == -2
            assert np.array_equal(pr, po), (chain, frame, level, np.nonzero(pr != po)[0][:6], pr[:4], po[:4])
            assert np.array_equal(ov_r[:n_ov], ov_o), (chain, frame, level, np.nonzero(ov_r[:n_ov] != ov_o)[0][:6])
            assert ps_r[0] == ps_o[0] == shape


def test_oracle_on_reference_made_chains(oracle):
    """tests/golden/imdct_ld_ref.npz (tools/make_golden_imdct_ld.py): needs no reference at run time"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from make_golden_imdct_ld import CHAINS, CONFIGS, FRAMES, chain_spec, crc, n_overlap
    gold = np.load(os.path.join(root, "tests", "golden", "imdct_ld_ref.npz"))
    of = oracle.lib.xo_imdct_ld_process
    of.restype = ctypes.c_int
    of.argtypes = [P32, P32, P16, ctypes.c_int, ctypes.c_int, ctypes.c_int, P16, ctypes.c_int]
    for g, (fl, eld) in enumerate(CONFIGS):
        for c in range(CHAINS):
            ov, ps = np.zeros(n_overlap(fl, eld), np.int32), np.zeros(1, np.int16)
            for f in range(FRAMES):
                spec, pcm = chain_spec(g, c, f), np.zeros(fl, np.int16)
                assert of(spec.ctypes.data_as(P32), ov.ctypes.data_as(P32), ps.ctypes.data_as(P16), int(gold["shape"][g, c, f]), fl, eld,
                          pcm.ctypes.data_as(P16), 1) == -2
                assert (crc(pcm), crc(ov)) == tuple(int(v) for v in gold["crc"][g, c, f]), (fl, eld, c, f)
            assert np.array_equal(pcm, gold["last"][g, c, :fl])
