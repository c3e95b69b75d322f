#!/usr/bin/env python3
"""Write tests/golden/streams/synth_*.aac: ADTS AAC-LC streams made by a syntax generator instead of an encoder, so that
the tools the reference's encoder never uses reach the parser tests -- pulse data, intensity stereo (both signs, with and
without M/S on the band), perceptual noise substitution (alone, correlated through M/S, in short blocks), every code book
with escapes up to 13 bits, TNS filters of every order in both directions on long and short windows, all window sequences
with random grouping, M/S mask modes 0 / 1 / 2, separate and common windows, fill and data stream elements in between.
The frames are random but legal; the code words come from libxaac_amd/host/tables_aac.inc (the books read out of the
reference's ROM).  Test infrastructure: the reference decoder itself (oracle/_ref) supplies the expected spectra / PCM
for these streams through tools/make_golden_parser.py like for the encoder-made ones."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "streams")
ZERO, ESC, NOISE, IS2, IS = 0, 11, 13, 14, 15


def tables():
    txt = open(os.path.join(ROOT, "libxaac_amd", "host", "tables_aac.inc")).read()
    t = {}
    for m in re.finditer(r"xh_(\w+)\[(\d+)\] = \{([^}]*)\}", txt):
        t[m.group(1)] = [int(v.replace("u", ""), 0) for v in m.group(3).replace("\n", " ").split(",") if v.strip()]
    books = []
    for cb in range(12):
        enc = {}
        for code, ln, idx in zip(t["hcb%d_code" % cb], t["hcb%d_len" % cb], t["hcb%d_idx" % cb]):
            enc[idx] = (code >> (32 - ln), ln)
        books.append(enc)
    return t, books


class Bits:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, value, bits):
        assert 0 <= value < (1 << bits) or bits == 0, (value, bits)
        self.v = (self.v << bits) | value
        self.n += bits

    def align(self):
        self.put(0, (8 - self.n % 8) % 8)

    def bytes(self):
        self.align()
        return self.v.to_bytes(self.n // 8, "big")


def adts(payload, sr_index, channels):
    n = len(payload) + 7
    h = Bits()
    h.put(0xfff, 12), h.put(0, 1), h.put(0, 2), h.put(1, 1)       # sync, MPEG-4, layer, no CRC
    h.put(1, 2), h.put(sr_index, 4), h.put(0, 1), h.put(channels, 3)  # AAC-LC
    h.put(0, 4), h.put(n, 13), h.put(0x7ff, 11), h.put(0, 2)
    return h.bytes() + payload


class Gen:
    def __init__(self, seed, sr_index, widths_long, widths_short, books, level):
        self.rng = np.random.default_rng(seed)
        self.sr_index, self.wl, self.ws, self.books, self.level = sr_index, widths_long, widths_short, books, level
        self.seq = [0, 0]

    def r(self, lo, hi):
        return int(self.rng.integers(lo, hi + 1))

    def next_sequence(self, prev):
        if prev in (0, 3):
            return 0 if self.r(0, 3) else 1
        return 2 if self.r(0, 2) == 0 else 3

    def ics(self, b, ch, forced=None):
        seq = self.next_sequence(self.seq[ch]) if forced is None else forced
        self.seq[ch] = seq
        shape = self.r(0, 1)
        b.put(0, 1), b.put(seq, 2), b.put(shape, 1)
        if seq != 2:
            max_sfb = self.r(0, len(self.wl)) if self.r(0, 9) == 0 else self.r(len(self.wl) // 2, len(self.wl))
            b.put(max_sfb, 6), b.put(0, 1)
            return dict(seq=seq, max_sfb=max_sfb, groups=[1])
        max_sfb = self.r(0, len(self.ws))
        grouping = self.r(0, 127)
        b.put(max_sfb, 4), b.put(grouping, 7)
        groups = [1]
        for i in range(7):
            if grouping & (0x40 >> i):
                groups[-1] += 1
            else:
                groups.append(1)
        return dict(seq=seq, max_sfb=max_sfb, groups=groups)

    def channel(self, b, ics, intensity):
        """what follows global_gain and ics_info in an individual_channel_stream"""
        long_block = ics["seq"] != 2
        widths = self.wl if long_block else self.ws
        max_sfb, groups = ics["max_sfb"], ics["groups"]
        # sections
        sect_bits, esc = (5, 31) if long_block else (3, 7)
        choices = [ZERO] + list(range(1, 12)) * 2 + [NOISE, NOISE] + ([IS, IS2, IS] if intensity else [])
        cbs = []
        for g in range(len(groups)):
            row, sfb = [], 0
            while sfb < max_sfb:
                cb = choices[self.r(0, len(choices) - 1)]
                ln = min(self.r(1, 8), max_sfb - sfb)
                b.put(cb, 4)
                left = ln
                while left >= esc:
                    b.put(esc, sect_bits)
                    left -= esc
                b.put(left, sect_bits)
                row += [cb] * ln
                sfb += ln
            cbs.append(row)
        # scale factors
        noise_seen = False
        for g in range(len(groups)):
            for cb in cbs[g]:
                if cb == ZERO:
                    continue
                if cb == NOISE and not noise_seen:
                    b.put(self.r(200, 330), 9)
                    noise_seen = True
                else:
                    d = self.r(-6, 6) if cb < NOISE else self.r(-4, 4)
                    code, ln = self.books[0][d + 60]
                    b.put(code, ln)
        # pulse data
        pulse = long_block and max_sfb > 0 and self.r(0, 2) == 0
        b.put(int(pulse), 1)
        if pulse:
            number = self.r(0, 3)
            start = self.r(0, min(max_sfb - 1, 30))
            b.put(number, 2), b.put(start, 6)
            for _ in range(number + 1):
                b.put(self.r(0, 31), 5), b.put(self.r(1, 15), 4)
        # TNS
        tns = max_sfb > 0 and self.r(0, 1) == 0
        b.put(int(tns), 1)
        if tns:
            for w in range(1 if long_block else 8):
                n_filt = self.r(0, 3 if long_block else 1) if self.r(0, 2) else 0
                b.put(n_filt, 2 if long_block else 1)
                if not n_filt:
                    continue
                res = self.r(0, 1)
                b.put(res, 1)
                for _ in range(n_filt):
                    b.put(self.r(0, (len(widths) if long_block else 8)), 6 if long_block else 4)   # length in bands
                    order = self.r(0, 12 if long_block else 7)
                    b.put(order, 5 if long_block else 3)
                    if order:
                        b.put(self.r(0, 1), 1)
                        compress = self.r(0, 1)
                        b.put(compress, 1)
                        bits = res + 3 - compress
                        for _ in range(order):
                            b.put(self.r(0, (1 << bits) - 1), bits)
        b.put(0, 1)  # gain_control_data_present
        # spectral data
        for g, glen in enumerate(groups):
            sfb = 0
            while sfb < max_sfb:
                cb = cbs[g][sfb]
                if cb == ZERO or cb >= NOISE:
                    sfb += 1
                    continue
                for _ in range(glen if not long_block else 1):
                    self.spectral(b, cb, widths[sfb])
                sfb += 1

    def spectral(self, b, cb, width):
        book = self.books[cb]
        if cb <= 4:
            for _ in range(width // 4):
                if cb <= 2:
                    v = [self.r(-1, 1) for _ in range(4)]
                    idx = 27 * (v[0] + 1) + 9 * (v[1] + 1) + 3 * (v[2] + 1) + v[3] + 1
                    b.put(*book[idx])
                else:
                    v = [self.r(-2, 2) for _ in range(4)]
                    idx = 27 * abs(v[0]) + 9 * abs(v[1]) + 3 * abs(v[2]) + abs(v[3])
                    b.put(*book[idx])
                    for x in v:
                        if x:
                            b.put(int(x < 0), 1)
        elif cb <= 10:
            lav = {5: 4, 6: 4, 7: 7, 8: 7, 9: 12, 10: 12}[cb]
            for _ in range(width // 2):
                v = [self.r(-lav, lav) if self.r(0, 3) else 0 for _ in range(2)]
                if cb <= 6:
                    b.put(*book[9 * (v[0] + 4) + v[1] + 4])
                else:
                    mod = 8 if cb <= 8 else 13
                    b.put(*book[mod * abs(v[0]) + abs(v[1])])
                    for x in v:
                        if x:
                            b.put(int(x < 0), 1)
        else:
            for _ in range(width // 2):
                v = []
                for _ in range(2):
                    k = self.r(0, 9)
                    m = self.r(0, 15) if k < 7 else (self.r(16, 200) if k < 9 else self.r(201, 8191))
                    v.append(-m if self.r(0, 1) else m)
                b.put(*book[17 * min(abs(v[0]), 16) + min(abs(v[1]), 16)])
                for x in v:
                    if x:
                        b.put(int(x < 0), 1)
                for x in v:
                    m = abs(x)
                    if m >= 16:
                        n = m.bit_length() - 5
                        b.put((1 << n) - 1, n), b.put(0, 1), b.put(m - (1 << (n + 4)), n + 4)

    def frame(self, channels):
        b = Bits()
        if self.r(0, 5) == 0:                       # a data stream element in front
            cnt, flag = self.r(0, 6), self.r(0, 1)
            b.put(4, 3), b.put(0, 4), b.put(flag, 1), b.put(cnt, 8)
            if flag:
                b.align()
            for _ in range(cnt):
                b.put(self.r(0, 255) if self.r(0, 3) else 0x12, 8)
        if channels == 1:
            b.put(0, 3), b.put(0, 4)
            self.channel_with_gain(b, None, False, False, 0)
            b.put(7, 3)
            return adts(b.bytes(), self.sr_index, 1)
        b.put(1, 3), b.put(0, 4)
        common = self.r(0, 3) != 0
        b.put(int(common), 1)
        if common:
            ics = self.ics(b, 0)
            self.seq[1] = self.seq[0]
            mask = self.r(0, 2)
            b.put(mask, 2)
            if mask == 1:
                for g in range(len(ics["groups"])):
                    for _ in range(ics["max_sfb"]):
                        b.put(self.r(0, 1), 1)
            for ch in range(2):   # intensity stereo needs the common window (its sign rides on the M/S mask)
                self.channel_with_gain(b, ics, ch == 1, True, ch)
        else:
            for ch in range(2):
                self.channel_with_gain(b, None, False, False, ch)
        if self.r(0, 3) == 0:                       # a fill element (not SBR)
            cnt = self.r(1, 10)
            b.put(6, 3), b.put(cnt, 4), b.put(0, 4), b.put(0, 4)
            for _ in range(cnt - 1):
                b.put(0xa5, 8)
        b.put(7, 3)
        return adts(b.bytes(), self.sr_index, 2)

    def channel_with_gain(self, b, ics, right, common, ch):
        """individual_channel_stream: global_gain, (ics_info), then the rest"""
        gain = self.r(100, 100 + self.level)
        b.put(gain, 8)
        if not common:
            ics = self.ics(b, ch)
        self.channel(b, ics, right)


def main():
    t, books = tables()
    wl = [w for w in t["sfb_48_1024"] if w > 0]
    ws = [w for w in t["sfb_48_128"] if w > 0]
    assert sum(wl) == 1024 and sum(ws) == 128
    for name, seed, frames, level, channels in (("synth_lc_a", 11, 96, 40, 2), ("synth_lc_b", 12, 96, 90, 2), ("synth_lc_mono", 13, 64, 60, 1)):
        g = Gen(seed, 3, wl, ws, books, level)
        data = b"".join(g.frame(channels) for _ in range(frames))
        open(os.path.join(OUT, name + ".aac"), "wb").write(data)
        print(name, frames, "frames", len(data), "bytes")


if __name__ == "__main__":
    main()
