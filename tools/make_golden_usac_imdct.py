#!/usr/bin/env python3
"""Build tests/golden/usac_imdct_ref.npz: chains of the REAL ixheaacd_fd_frm_dec (decoder/ixheaacd_imdct.c:596; ccfl 1024
and 768, FD after FD, no FAC) run by the compiled reference (oracle/_ref/libref_harness.so through oracle/ref_usac_adapter.c) with
the overlap carried from frame to frame along legal window-sequence walks.

The 1024 spectral lines of a frame are NOT stored: tests regenerate them from (chain, frame) with chain_coef() below
(integer arithmetic on a counter, no library RNG).  Stored per frame: window sequence, shape, CRC32 of the reference's
Q15 output and of its overlap after the call; the last frame's output and overlap in full.  Data only; runs only where
/root/reference is."""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

CHAINS, FRAMES = 24, 44
NEXT = {0: (0, 1), 3: (0, 1), 1: (2, 3, 4), 2: (2, 3, 4), 4: (2, 3, 4)}


def _mix(base, n):
    z = (np.uint64(base) * np.uint64(4096) + np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


CHAINS_768, FRAMES_768 = 16, 40


def chain_coef(chain, frame, ccfl=1024):
    """ccfl spectral lines: uniform noise at a level that walks with (chain, frame); every 5th frame sparse tonal lines,
    every 11th silence, every 13th one full-scale line in low-level noise (ccfl 768: its own counter range)"""
    z = _mix((1 << 30) | (chain << 12) | frame | ((1 << 24) if ccfl == 768 else 0), ccfl)
    v = (z >> np.uint64(32)).astype(np.int64) - (1 << 31)            # 32-bit signed, uniform
    level = (3 * chain + 5 * frame) % 27                             # right shift 0 .. 26
    x = v >> level
    if frame % 11 == 10:
        x[:] = 0
    elif frame % 5 == 4:
        keep = (z & np.uint64(63)) == 0
        x = np.where(keep, x, 0)
    elif frame % 13 == 12:
        x = v >> 25
        x[int(z[0] & np.uint64(1023)) % ccfl] = -(1 << 31) if (int(z[1]) & 1) else (1 << 31) - 1
    return x.astype(np.int32)


def chain_side(chain, frame, seq):
    """(shape, next window sequence) of the frame"""
    z = _mix((2 << 30) | (chain << 12) | frame, 2)
    nxt = NEXT[seq]
    return int(z[0] & np.uint64(1)), int(nxt[int(z[1] >> np.uint64(8)) % len(nxt)])


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def main():
    import oracle_lib
    import test_usac_oracle_vs_reference as t
    ref = oracle_lib.load_reference()
    side = np.zeros((CHAINS, FRAMES, 2), np.uint8)
    crcs = np.zeros((CHAINS, FRAMES, 2), np.uint32)
    last = np.zeros((CHAINS, 2, 1024), np.int32)
    for c in range(CHAINS):
        ov = np.zeros(1024, np.int32)
        seq, shape_prev = (0, 1, 3, 4)[c % 4] if c >= 4 else 0, c & 1
        for f in range(FRAMES):
            shape, nxt = chain_side(c, f, seq)
            rc, _, ov, out, _ = t.ref_call(ref, chain_coef(c, f), ov, seq, shape, shape_prev)
            assert rc == 0
            side[c, f] = seq, shape
            crcs[c, f] = crc(out), crc(ov)
            seq, shape_prev = nxt, shape
        last[c, 0], last[c, 1] = out, ov
    # ccfl 768 (768- / 96-line transforms = 3 x radix-4 + ixheaacd_complex_fft_p3's three-point stage, the 768 / 96 windows)
    side8 = np.zeros((CHAINS_768, FRAMES_768, 2), np.uint8)
    crcs8 = np.zeros((CHAINS_768, FRAMES_768, 2), np.uint32)
    last8 = np.zeros((CHAINS_768, 2, 768), np.int32)
    for c in range(CHAINS_768):
        ov = np.zeros(768, np.int32)
        seq, shape_prev = (0, 1, 3, 4)[c % 4] if c >= 4 else 0, c & 1
        for f in range(FRAMES_768):
            shape, nxt = chain_side(c + 64, f, seq)
            rc, _, ov, out, _ = t.ref_call(ref, chain_coef(c, f, 768), ov, seq, shape, shape_prev)
            assert rc == 0
            side8[c, f] = seq, shape
            crcs8[c, f] = crc(out), crc(ov)
            seq, shape_prev = nxt, shape
        last8[c, 0], last8[c, 1] = out, ov
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "usac_imdct_ref.npz"), side=side, crc=crcs, last=last, side768=side8,
                        crc768=crcs8, last768=last8)
    print("wrote", CHAINS, "chains x", FRAMES, "frames; sequences seen:", np.bincount(side[:, :, 0].ravel(), minlength=5),
          "and", CHAINS_768, "x", FRAMES_768, "of 768 lines:", np.bincount(side8[:, :, 0].ravel(), minlength=5))
    make_lpd(ref, t)


LPD_CHAINS, LPD_FRAMES = 16, 34     # per ccfl: 544 frames, about a third of them behind an LPD frame


def lpd_plan(ccfl, chain, frame, seq):
    """(window sequence, shape, td_frame_prev, fac_data_present, next sequence) of a frame of the LPD walks: every third frame
    or so follows an LPD frame (then it opens with a short slope and the previous shape counts as sine), two in three of
    those carry FAC data"""
    z = _mix((3 << 30) | ((ccfl == 768) << 24) | (chain << 12) | frame, 4)
    td = int(z[0] >> np.uint64(11)) % 3 == 0 and frame > 0
    if td:
        seq = (2, 3, 4)[int(z[1] >> np.uint64(9)) % 3]
    fac = td and int(z[2] >> np.uint64(13)) % 3 != 0
    nxt = NEXT[seq]
    return seq, int(z[3] & np.uint64(1)), int(td), int(fac), int(nxt[int(z[3] >> np.uint64(8)) % len(nxt)])


def make_lpd(ref, t):
    """tests/golden/usac_lpd_ref.npz: the REAL ixheaacd_fd_frm_dec on walks with LPD -> FD transitions.  The FAC signal of a
    frame is the reference's own ixheaacd_cal_fac_data output for LPD-side inputs drawn here (they are not stored; the
    signal and its exponent are: the boundary takes exactly those from the host)."""
    rng = np.random.default_rng(77)
    d = {}
    for ccfl in (1024, 768):
        side = np.zeros((LPD_CHAINS, LPD_FRAMES, 4), np.uint8)
        crcs = np.zeros((LPD_CHAINS, LPD_FRAMES, 2), np.uint32)
        facq = np.zeros((LPD_CHAINS, LPD_FRAMES), np.int32)
        facs = np.zeros((LPD_CHAINS, LPD_FRAMES, 256), np.int32)
        for c in range(LPD_CHAINS):
            ov = np.zeros(ccfl, np.int32)
            seq, shape_prev = 0, c & 1
            for f in range(LPD_FRAMES):
                seq, shape, td, fac, nxt = lpd_plan(ccfl, c, f, seq)
                if td:
                    shape_prev = 0
                rc, _, ov, out, sig, q = t.ref_call_lpd(ref, chain_coef(c + 32, f, ccfl), ov, seq, shape, shape_prev, td, fac,
                                                        t.lpd_side(rng, ccfl, seq, fac))
                assert rc == 0, (ccfl, c, f, rc)
                side[c, f] = seq, shape, td, fac
                crcs[c, f] = crc(out), crc(ov)
                if fac:
                    facq[c, f], facs[c, f] = q, sig
                seq, shape_prev = nxt, shape
        k = str(ccfl)
        d.update({"side" + k: side, "crc" + k: crcs, "fac_q" + k: facq, "fac" + k: facs})
        print("ccfl", ccfl, ":", LPD_CHAINS * LPD_FRAMES, "frames,", int(side[:, :, 2].sum()), "behind an LPD frame,", int(side[:, :, 3].sum()), "with FAC")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "usac_lpd_ref.npz"), **d)


if __name__ == "__main__":
    main()
