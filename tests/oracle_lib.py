"""ctypes access to the checkers: oracle/liboracle.so (our CPU restatement) and,
where present, oracle/_ref/libref_harness.so (the real reference).  Test
infrastructure only -- nothing in the product imports this."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P32 = ctypes.POINTER(ctypes.c_int32)
P16 = ctypes.POINTER(ctypes.c_int16)
P8 = ctypes.POINTER(ctypes.c_int8)
PU8 = ctypes.POINTER(ctypes.c_uint8)


def _p(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.xo_imdct_process.restype = ctypes.c_int
        lib.xo_imdct_process.argtypes = [P32, P32, P16, P16, ctypes.c_int, ctypes.c_int, P32, ctypes.c_int]
        lib.xo_imdct_batch.restype = None
        lib.xo_imdct_batch.argtypes = [ctypes.c_int, P32, P32, P16, P16, PU8, PU8, P32, P16, P8, ctypes.c_int]
        lib.xo_pcm16.restype = None
        lib.xo_pcm16.argtypes = [P32, ctypes.c_int, P16, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]

    def imdct_process(self, spec, ovl, prev_seq, prev_shape, seq, shape):
        """one channel-frame -> (qshift_adj, out[1024], new_ovl[512], new_seq, new_shape)"""
        spec = np.ascontiguousarray(spec, np.int32)
        ovl = np.array(ovl, np.int32)
        ps = np.array([prev_seq], np.int16)
        psh = np.array([prev_shape], np.int16)
        out = np.zeros(1024, np.int32)
        q = self.lib.xo_imdct_process(_p(spec, P32), _p(ovl, P32), _p(ps, P16), _p(psh, P16), int(seq), int(shape),
                                      _p(out, P32), 1)
        return q, out, ovl, int(ps[0]), int(psh[0])

    def imdct_batch(self, spec, ics, ovl, state, want_out32=True, want_pcm=True, pcm_mode=0, ch_fac=1):
        """batch with the C-ABI's conventions: ics/state uint8[N,2]; returns dict of new arrays.
        Outputs are interleaved at stride ch_fac like the product's."""
        n = spec.shape[0]
        spec = np.ascontiguousarray(spec, np.int32)
        ovl = np.array(ovl, np.int32)
        pseq = np.ascontiguousarray(state[:, 0].astype(np.int16))
        pshape = np.ascontiguousarray(state[:, 1].astype(np.int16))
        seq = np.ascontiguousarray(ics[:, 0].astype(np.uint8))
        shape = np.ascontiguousarray(ics[:, 1].astype(np.uint8))
        out32 = np.zeros((n, 1024), np.int32)
        pcm = np.zeros((n, 1024), np.int16)
        qadj = np.zeros(n, np.int8)
        self.lib.xo_imdct_batch(n, _p(spec, P32), _p(ovl, P32), _p(pseq, P16), _p(pshape, P16), _p(seq, PU8),
                                _p(shape, PU8), _p(out32, P32), _p(pcm, P16), _p(qadj, P8), int(pcm_mode))
        if ch_fac != 1:
            out32 = out32.reshape(n // ch_fac, ch_fac, 1024).transpose(0, 2, 1).reshape(n, 1024)
            # the interleaved block's PCM16 is made IN PLACE like the reference makes it (xo_pcm16_block): for
            # stereo + the SBR hand-off that is not the per-channel conversion (oracle_imdct.c)
            blk = np.ascontiguousarray(out32).copy()
            pcm = np.zeros((n, 1024), np.int16)
            self.lib.xo_pcm16_block.restype = None
            for au in range(n // ch_fac):
                self.lib.xo_pcm16_block(blk[au * ch_fac:].ctypes.data_as(ctypes.c_void_p),
                                        np.ascontiguousarray(qadj[au * ch_fac:(au + 1) * ch_fac]).ctypes.data_as(ctypes.c_void_p),
                                        ch_fac, int(pcm_mode), pcm[au * ch_fac:].ctypes.data_as(ctypes.c_void_p))
        return {"out32": np.ascontiguousarray(out32), "pcm16": np.ascontiguousarray(pcm), "qshift_adj": qadj,
                "overlap": ovl, "state": np.stack([pseq, pshape], 1).astype(np.uint8)}


class QmfAnaState(ctypes.Structure):
    _fields_ = [("ring", ctypes.c_int16 * 320), ("wr", ctypes.c_int16), ("phase", ctypes.c_int16)]


class QmfSynState(ctypes.Structure):
    _fields_ = [("ring", ctypes.c_int16 * 1280), ("drc_offset", ctypes.c_int16), ("phase", ctypes.c_int16)]


def qmf_analysis_batch(orc, pcm, state, low_pow, usb, slot_stride, ch_fac=1):
    """oracle over a batch with the C ABI's conventions; state int16[n,322] (updated copy returned)"""
    n = state.shape[0]
    state = np.array(state, np.int16)
    qmf = np.zeros((n, 32, slot_stride), np.int32)
    for i in range(n):
        st = QmfAnaState.from_buffer(state[i])
        src = pcm[(i // ch_fac) * 1024 * ch_fac + (i % ch_fac):]
        orc.lib.xo_qmf_analysis(_p(np.ascontiguousarray(src), P16), ch_fac, ctypes.byref(st), int(low_pow), int(usb),
                                _p(qmf[i], P32), slot_stride)
    return qmf, state


def qmf_synthesis_batch(orc, qmf, scale, state, low_pow, lsb, usb, split, ch_fac=1):
    n = state.shape[0]
    state = np.array(state, np.int16)
    slot_stride = qmf.shape[2]
    pcm = np.zeros(n * 2048, np.int16)
    for i in range(n):
        st = QmfSynState.from_buffer(state[i])
        dst = pcm[(i // ch_fac) * 2048 * ch_fac + (i % ch_fac):]
        orc.lib.xo_qmf_synthesis(_p(np.ascontiguousarray(qmf[i]), P32), slot_stride,
                                 _p(np.ascontiguousarray(scale[i]), P16), int(lsb), int(usb), int(split),
                                 ctypes.byref(st), int(low_pow), _p(dst, P16), ch_fac)
    return pcm, state


class Reference:
    def __init__(self, lib):
        self.lib = lib
        lib.ref_imdct_process.restype = ctypes.c_int
        lib.ref_imdct_process.argtypes = [P32, P32, P16, P16, ctypes.c_int, ctypes.c_int, P32, ctypes.c_int]

    def imdct_process(self, spec, ovl, prev_seq, prev_shape, seq, shape):
        spec = np.array(spec, np.int32)  # the reference clobbers its input
        ovl = np.array(ovl, np.int32)
        ps = np.array([prev_seq], np.int16)
        psh = np.array([prev_shape], np.int16)
        out = np.zeros(1024, np.int32)
        q = self.lib.ref_imdct_process(_p(spec, P32), _p(ovl, P32), _p(ps, P16), _p(psh, P16), int(seq), int(shape),
                                       _p(out, P32), 1)
        return q, out, ovl, int(ps[0]), int(psh[0])


def load_oracle():
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return Oracle(ctypes.CDLL(so))


def load_reference():
    so = os.path.join(ROOT, "oracle", "_ref", "libref_harness.so")
    if not os.path.exists(so):
        return None
    try:
        return Reference(ctypes.CDLL(so))
    except OSError:
        return None


def random_case(rng, n, mag=None, ovl_mag=None):
    """n channel-frames of seeded synthetic input covering the block-exponent range"""
    spec = np.zeros((n, 1024), np.int32)
    ovl = np.zeros((n, 512), np.int32)
    for i in range(n):
        m = int(rng.integers(1, 31)) if mag is None else mag
        spec[i] = rng.integers(-(1 << m), 1 << m, 1024)
        k = int(rng.integers(0, 8))
        if k == 0:
            spec[i, rng.integers(0, 1024, 900)] = 0
        elif k == 1:
            spec[i, 640:] = 0
        om = (int(rng.integers(1, 31)) if rng.integers(0, 3) == 0 else 15) if ovl_mag is None else ovl_mag
        ovl[i] = rng.integers(-(1 << om), 1 << om, 512)
    return spec, ovl
