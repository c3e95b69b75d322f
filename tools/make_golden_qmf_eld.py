#!/usr/bin/env python3
"""Build tests/golden/qmf_eld_ref.npz: chains of the REAL LD / ELD complex QMF banks -- ixheaacd_cplx_anal_qmffilt and
ixheaacd_cplx_synt_qmffilt with AOT_ER_AAC_ELD (generic/ixheaacd_qmf_dec_generic.c:590-741, ixheaacd_qmf_dec.c:811-1135)
driven through oracle/_ref/libref_harness.so (oracle/ref_harness.c: ref_qmf_analysis_eld / ref_qmf_synthesis_eld), frames
of 16 and of 15 slots, ring and pointer state carried by the reference-side arrays from frame to frame.  Inputs are not
stored: eld_pcm() / eld_qmf() regenerate them from (kind, chain, frame) with an integer counter generator.  Stored per
frame: the call's parameters and CRC32s of the output, the ring and the four state words after the call.
Data only; runs only where /root/reference exists."""
import ctypes
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P16 = ctypes.POINTER(ctypes.c_int16)
P32 = ctypes.POINTER(ctypes.c_int32)
CHAINS, FRAMES = 10, 26


def _mix(base, n):
    z = (np.uint64(base) * np.uint64(1 << 20) + np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def eld_pcm(n_slots, chain, frame):
    """32 n_slots core samples: levels from full scale to a few LSB, every seventh frame a constant rail"""
    v = (_mix((1 << 30) | (n_slots << 20) | (chain << 10) | frame, 32 * n_slots) >> np.uint64(40)).astype(np.int64) - (1 << 23)
    amp = (32767, 3000, 12, 32767, 700)[(chain + frame) % 5]
    x = ((v * amp) >> 23).astype(np.int16)
    if frame % 7 == 3:
        x[:] = 32767 if (chain + frame) % 2 else -32768
    return x


def eld_qmf(n_slots, chain, frame):
    """n_slots rows of 64 re | 64 im int32 subband samples at a level between 2^8 and 2^31 (clipping included)"""
    z = _mix((2 << 30) | (n_slots << 20) | (chain << 10) | frame, 128 * n_slots)
    v = (z >> np.uint64(32)).astype(np.int64) - (1 << 31)
    sh = (23, 2, 12, 0, 17, 7)[(chain + 3 * frame) % 6]
    q = (v >> sh).astype(np.int32).reshape(n_slots, 128)
    if frame % 6 == 5:
        q[:] = 2 ** 31 - 1 if (chain + frame) % 2 else -2 ** 31
    return q


def params(n_slots, chain, frame):
    z = _mix((3 << 30) | (n_slots << 20) | (chain << 10) | frame, 8) >> np.uint64(33)
    usb_a = int(z[0] % 33)
    sf = np.array([int(z[1] % 16) - 12, int(z[2] % 16) - 12, int(z[3] % 16) - 12, int(z[4] % 12) - 10], np.int16)
    lsb = int(z[5] % 40)
    usb = lsb + int(z[6] % (65 - lsb))
    split = int(z[7] % (n_slots + 1))
    return usb_a, sf, lsb, usb, split


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def main():
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_harness.so"))
    ana, syn = ref.ref_qmf_analysis_eld, ref.ref_qmf_synthesis_eld
    ana.restype = syn.restype = None
    ana.argtypes = [P16, ctypes.c_int, P16, P16, ctypes.c_int, ctypes.c_int, P32, ctypes.c_int]
    syn.argtypes = [P32, ctypes.c_int, P16, ctypes.c_int, ctypes.c_int, ctypes.c_int, P16, P16, ctypes.c_int, P16, ctypes.c_int]
    d = {}
    for n_slots in (16, 15):
        ca = np.zeros((CHAINS, FRAMES, 3), np.uint32)
        cs = np.zeros((CHAINS, FRAMES, 3), np.uint32)
        for c in range(CHAINS):
            ring_a, st_a = np.zeros(320, np.int16), np.array([0, 0, 32, 0], np.int16)
            ring_s, st_s = np.zeros(1280, np.int16), np.array([0, 0, 0, 64], np.int16)
            for f in range(FRAMES):
                usb_a, sf, lsb, usb, split = params(n_slots, c, f)
                pcm = eld_pcm(n_slots, c, f)
                q = np.full((n_slots, 128), 5, np.int32)
                ana(pcm.ctypes.data_as(P16), 1, ring_a.ctypes.data_as(P16), st_a.ctypes.data_as(P16), n_slots, usb_a, q.ctypes.data_as(P32), 128)
                ca[c, f] = crc(q), crc(ring_a), crc(st_a)
                qq = eld_qmf(n_slots, c, f).copy()
                out = np.zeros(64 * n_slots, np.int16)
                syn(qq.ctypes.data_as(P32), 128, sf.ctypes.data_as(P16), lsb, usb, split, ring_s.ctypes.data_as(P16), st_s.ctypes.data_as(P16),
                    n_slots, out.ctypes.data_as(P16), 1)
                cs[c, f] = crc(out), crc(ring_s), crc(st_s)
        d["ana_crc_%d" % n_slots] = ca
        d["syn_crc_%d" % n_slots] = cs
    dst = os.path.join(ROOT, "tests", "golden", "qmf_eld_ref.npz")
    np.savez_compressed(dst, **d)
    print(dst, os.path.getsize(dst), "bytes;", 2 * CHAINS * FRAMES, "frames of each bank")


if __name__ == "__main__":
    main()
