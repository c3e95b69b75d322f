"""The 960-line IMDCT restatement (libxaac_amd/csrc/imdct960.h through oracle/oracle_imdct960.cpp) against the compiled
reference's ixheaacd_imdct_process with frame_length 960: every (previous, current) window sequence pair with both window
shapes on either side, levels from silence to full scale, legal random walks with the overlap carried."""
import ctypes

import numpy as np
import pytest

P32 = ctypes.POINTER(ctypes.c_int32)
P16 = ctypes.POINTER(ctypes.c_int16)
LEGAL_NEXT = {0: (0, 1), 1: (2, 3), 2: (2, 3), 3: (0, 1)}


def bind(lib, name):
    fn = getattr(lib, name)
    fn.restype = ctypes.c_int
    fn.argtypes = [P32, P32, P16, P16, ctypes.c_int, ctypes.c_int, P32, ctypes.c_int]
    return fn


def spectrum(rng, level, kind):
    if kind == 0:
        x = rng.integers(-level, level + 1, 960)
    elif kind == 1:
        x = np.zeros(960, np.int64)
        x[rng.integers(0, 960, 6)] = rng.integers(-level, level + 1, 6)
    else:
        x = (rng.standard_normal(960) * level / 3).clip(-2 ** 31, 2 ** 31 - 1)
    return x.astype(np.int32)


def run_pair(rf, of, spec, ovl, pseq, pshape, seq, shape):
    outs = []
    for fn, n_spec, n_ovl in ((rf, 1024, 512), (of, 960, 480)):
        s = np.zeros(n_spec, np.int32)
        s[:960] = spec
        o = np.zeros(n_ovl, np.int32)
        o[:480] = ovl
        ps, pw = np.array([pseq], np.int16), np.array([pshape], np.int16)
        out = np.zeros(960, np.int32)
        q = fn(s.ctypes.data_as(P32), o.ctypes.data_as(P32), ps.ctypes.data_as(P16), pw.ctypes.data_as(P16), seq, shape,
               out.ctypes.data_as(P32), 1)
        outs.append((q, out, o[:480].copy(), int(ps[0]), int(pw[0])))
    return outs


def test_every_transition_every_level(oracle, reference):
    rf, of = bind(reference.lib, "ref_imdct960_process"), bind(oracle.lib, "xo_imdct960_process")
    rng = np.random.default_rng(960)
    n = 0
    for pseq in range(4):
        for seq in range(4):
            for pshape in range(2):
                for shape in range(2):
                    for level in (0, 1, 300, 2 ** 17, 2 ** 24, 2 ** 30, 2 ** 31 - 1):
                        spec = spectrum(rng, level, n % 3)
                        if level == 2 ** 31 - 1 and n % 2:
                            spec[:] = -2 ** 31
                        ovl = rng.integers(-2 ** (15 + n % 3), 2 ** (15 + n % 3), 480).astype(np.int32)
                        (qr, outr, ovr, sr, wr), (qo, outo, ovo, so, wo) = run_pair(rf, of, spec, ovl, pseq, pshape, seq, shape)
                        assert qr == qo, (pseq, seq, level)
                        assert np.array_equal(outr, outo), (pseq, seq, pshape, shape, level, np.nonzero(outr != outo)[0][:6])
                        assert np.array_equal(ovr, ovo), (pseq, seq, pshape, shape, level, np.nonzero(ovr != ovo)[0][:6])
                        assert (sr, wr) == (so, wo) == (seq, shape)
                        n += 1
    assert n == 4 * 4 * 4 * 7


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_legal_walk_with_state(oracle, reference, seed):
    rf, of = bind(reference.lib, "ref_imdct960_process"), bind(oracle.lib, "xo_imdct960_process")
    rng = np.random.default_rng(9600 + seed)
    ovl_r, ovl_o = np.zeros(512, np.int32), np.zeros(480, np.int32)
    st_r, st_o = [np.zeros(1, np.int16), np.zeros(1, np.int16)], [np.zeros(1, np.int16), np.zeros(1, np.int16)]
    seq = 0
    for frame in range(60):
        seq = int(rng.choice(LEGAL_NEXT[seq]))
        shape = int(rng.integers(0, 2))
        spec = spectrum(rng, int(2 ** rng.integers(4, 31)), frame % 3)
        sr = np.zeros(1024, np.int32)
        sr[:960] = spec
        outr, outo = np.zeros(960, np.int32), np.zeros(960, np.int32)
        qr = rf(sr.ctypes.data_as(P32), ovl_r.ctypes.data_as(P32), st_r[0].ctypes.data_as(P16), st_r[1].ctypes.data_as(P16), seq, shape,
                outr.ctypes.data_as(P32), 1)
        so = spec.copy()
        qo = of(so.ctypes.data_as(P32), ovl_o.ctypes.data_as(P32), st_o[0].ctypes.data_as(P16), st_o[1].ctypes.data_as(P16), seq, shape,
                outo.ctypes.data_as(P32), 1)
        assert np.array_equal(so, spec)
        assert qr == qo and np.array_equal(outr, outo), (frame, seq)
        assert np.array_equal(ovl_r[:480], ovl_o), (frame, seq)


def test_oracle_on_reference_made_chains(oracle):
    """tests/golden/imdct960_ref.npz (tools/make_golden_imdct960.py: the compiled reference along legal walks): the
    restatement reproduces every frame's output and overlap CRC; this one needs no reference at run time"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from make_golden_imdct960 import CHAINS, FRAMES, chain_spec, crc
    gold = np.load(os.path.join(root, "tests", "golden", "imdct960_ref.npz"))
    of = bind(oracle.lib, "xo_imdct960_process")
    for c in range(CHAINS):
        ov = np.zeros(480, np.int32)
        ps, pw = np.zeros(1, np.int16), np.zeros(1, np.int16)
        for f in range(FRAMES):
            seq, shape, q = (int(v) for v in gold["side"][c, f])
            spec = chain_spec(c, f)
            out = np.zeros(960, np.int32)
            got = of(spec.ctypes.data_as(P32), ov.ctypes.data_as(P32), ps.ctypes.data_as(P16), pw.ctypes.data_as(P16), seq, shape,
                     out.ctypes.data_as(P32), 1)
            assert got == q and (crc(out), crc(ov)) == tuple(int(v) for v in gold["crc"][c, f]), (c, f, seq)
        assert np.array_equal(out, gold["last"][c, 0]) and np.array_equal(ov, gold["last"][c, 1, :480])
