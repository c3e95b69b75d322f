"""ctypes mirrors of include/xaac_pvc.h and the seeded frame chains the PVC tests and tools/make_golden_pvc.py share."""
import ctypes
import zlib

import numpy as np


class PvcFrame(ctypes.Structure):
    _fields_ = [("pvc_mode", ctypes.c_uint8), ("ns_mode", ctypes.c_uint8), ("pvc_rate", ctypes.c_uint8), ("low_power", ctypes.c_uint8),
                ("first_bnd_idx", ctypes.c_int16), ("first_pvc_timeslot", ctypes.c_int16), ("pvc_id", ctypes.c_uint16 * 16)]


class PvcState(ctypes.Structure):
    _fields_ = [("esg", (ctypes.c_float * 3) * 15), ("prev_first_bnd_idx", ctypes.c_int16), ("prev_pvc_id", ctypes.c_uint16),
                ("prev_pvc_flg", ctypes.c_uint8), ("prev_pvc_rate", ctypes.c_uint8), ("reserved", ctypes.c_uint8 * 2)]


PF = ctypes.POINTER(ctypes.c_float)


def bind(lib, name):
    fn = getattr(lib, name)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, PF, PF, ctypes.c_void_p, PF]
    return fn


def chain(seed, n_frames):
    """One channel's frames: (PvcFrame, re[64, 64], im[64, 64], host_clears_flag).  A stream keeps its rate and mostly its
    mode and start band; now and then the band moves or a frame without PVC lies in between (prev_pvc_flg cleared by the
    host, sbr_dec.c:948): both restart the history.  Levels from digital silence to 2^20."""
    rng = np.random.default_rng(seed)
    rate = int(rng.choice([2, 2, 2, 4]))
    mode = int(rng.integers(1, 3))
    first = int(rng.integers(0, 17 if rate == 4 else 33))
    out = []
    for n in range(n_frames):
        if rng.integers(0, 6) == 0:
            first = int(np.clip(first + rng.integers(-3, 4), 0, 16 if rate == 4 else 32))
        if rng.integers(0, 9) == 0:
            mode = 3 - mode
        f = PvcFrame()
        f.pvc_mode, f.ns_mode, f.pvc_rate, f.low_power = mode, int(rng.integers(0, 2)), rate, int(rng.integers(0, 5) == 0)
        f.first_bnd_idx, f.first_pvc_timeslot = first, int(rng.choice([0, 0, 0, 1, 3, 7, 15]))
        ids = rng.integers(0, 128, 16)
        if rng.integers(0, 2):
            ids[:] = ids[0]          # one code book entry per frame is the usual payload
        for t in range(16):
            f.pvc_id[t] = int(ids[t])
        level = np.float32(2.0 ** rng.integers(-8, 21))
        re = (rng.standard_normal((64, 64)) * level).astype(np.float32)
        im = (rng.standard_normal((64, 64)) * level).astype(np.float32)
        kind = rng.integers(0, 8)
        if kind == 0:
            re[:], im[:] = 0, 0      # silence: the -10 dB floor
        elif kind == 1:
            re[rng.integers(0, 64):] = 0
            im[:] = 0
        elif kind == 2:
            re[:, ::2] *= np.float32(1e-4)
        out.append((f, re, im, bool(n > 0 and rng.integers(0, 10) == 0)))
    return out


def crc(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(bytes(a), c)
    return c & 0xffffffff


def walk(fn, frames):
    """run fn over a chain with carried state -> uint32[n, 3]: return code, CRC of the 1024 output floats, CRC of the state"""
    st = PvcState()
    res = np.zeros((len(frames), 3), np.uint32)
    for n, (f, re, im, clear) in enumerate(frames):
        if clear:
            st.prev_pvc_flg = 0
        out = np.full((16, 64), np.float32(-7.0))
        rc = fn(ctypes.byref(f), re.ctypes.data_as(PF), im.ctypes.data_as(PF), ctypes.byref(st), out.ctypes.data_as(PF))
        res[n] = (rc & 0xffffffff, crc(out), crc(st))
    return res
