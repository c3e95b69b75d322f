#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Recompute tools/table_fixups.json: the sparse list of
entries where the reference ROM (read from the compiled reference,
oracle/_ref/libxaacdec_ref.so, symbol ixheaacd_imdct_tables laid out as
decoder/ixheaacd_aac_rom.h:112-121) differs from the closed-form tables of
tools/gen_tables.py.  Only runs where oracle/_ref exists (this container)."""
import ctypes
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tools"))
import gen_tables  # noqa: E402


class ImdctRom(ctypes.Structure):
    _fields_ = [("pre_cs", ctypes.c_int16 * 514), ("digrev_long", ctypes.c_uint8 * 64),
                ("digrev_short", ctypes.c_uint8 * 8), ("fft_tw", ctypes.c_int32 * 448),
                ("win_long_sine", ctypes.c_int16 * 1024), ("win_long_kbd", ctypes.c_int16 * 1024),
                ("win_short_sine", ctypes.c_int16 * 128), ("win_short_kbd", ctypes.c_int16 * 128)]


def reference_tables(so=os.path.join(HERE, "_ref", "libxaacdec_ref.so")):
    lib = ctypes.CDLL(so)
    rom = ImdctRom.in_dll(lib, "ixheaacd_imdct_tables")
    t = {n: np.array(getattr(rom, n)[:], dtype=np.int64) for n, _ in ImdctRom._fields_}
    tw = t.pop("fft_tw")
    t["fft_tw_hi"] = tw >> 16
    t["fft_tw_lo"] = ((tw & 0xFFFF) ^ 0x8000) - 0x8000
    return t


if __name__ == "__main__":
    ref = reference_tables()
    mine = gen_tables.formula_tables()
    fix = {}
    for name, want in ref.items():
        got = mine[name]
        assert got.shape == want.shape, name
        bad = np.nonzero(got != want)[0]
        if len(bad):
            fix[name] = {str(int(i)): int(want[i]) for i in bad}
        print("%-16s %4d entries, %3d fixups, max |delta| %d" % (
            name, len(want), len(bad), int(np.max(np.abs(got - want))) if len(bad) else 0))
    with open(gen_tables.FIXUPS, "w") as f:
        json.dump(fix, f, indent=1, sort_keys=True)
