#!/usr/bin/env python3
"""Developer tool (GPU box): the fuzzed-grid chains of tests/test_sbr_gpu.py / test_sbr_hq_gpu.py (envelope grids of every
kind, moving band limit, state on the device, every frame against the oracle) for many seeds:
    python tools/soak_sbr_fuzz.py [first_seed] [seeds]
prints one line per mismatch and a summary."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import sbr_capture as cap
    import oracle_lib
    import libxaac_amd
    import test_sbr_gpu as LP
    import test_sbr_hq_gpu as HQ
    from test_env_pairs_cpu import _fuzz_frame
    first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 1000), (int(sys.argv[2]) if len(sys.argv) > 2 else 8)
    orc = oracle_lib.load_oracle()
    ctx = libxaac_amd.XaacContext(0, 0)
    P16 = ctypes.POINTER(ctypes.c_int16)
    bad = frames = 0
    for seed in range(first, first + count):
        for mode in ("lp", "hq"):
            recs = cap.read_records(os.path.join(ROOT, "tests", "golden", "sbr_lp_records.bin.gz" if mode == "lp" else "sbr_hq_ps_records.bin.gz"))
            n = len(recs)
            rng = np.random.default_rng(seed)
            st = [cap.State.from_buffer_copy(bytes(r["st0"])) for r in recs]
            ps = [cap.PsState.from_buffer_copy(bytes(r["ps0"])) for r in recs] if mode == "hq" else None
            hdrs = [cap.Header.from_buffer_copy(bytes(r["header"])) for r in recs]
            pfs = [cap.PsFrame.from_buffer_copy(bytes(r["ps_frame"])) for r in recs] if mode == "hq" else None
            for step in range(10):
                fr = []
                for i, r in enumerate(recs):
                    f = cap.Frame.from_buffer_copy(bytes(r["frame"]))
                    _fuzz_frame(rng, hdrs[i], f, int(rng.integers(0, 3)))
                    if rng.integers(0, 3) == 0:
                        f.max_qmf_subband_aac = int(np.clip(f.max_qmf_subband_aac + rng.integers(-6, 7), hdrs[i].sub_band_start, 32))
                    if rng.integers(0, 8) == 0:
                        hdrs[i].interpol_freq = 1 - hdrs[i].interpol_freq
                    fr.append(f)
                amp = [30000, 3000, 200, 12, 0][int(rng.integers(0, 5))]
                pcm = rng.integers(-amp, amp + 1, (n, 1024)).astype(np.int16)
                if mode == "lp":
                    out, stb, status = LP.gpu_run(ctx, hdrs, fr, st, pcm.reshape(-1))
                    psb = None
                else:
                    out, stb, psb, status = HQ.gpu_run(ctx, hdrs, fr, st, pfs, ps, pcm.reshape(-1))
                per = 2048 if mode == "lp" else 4096
                for i in range(n):
                    so = cap.State.from_buffer_copy(bytes(st[i]))
                    want = np.zeros(per, np.int16)
                    if mode == "lp":
                        rc = orc.lib.xo_sbr_dec_lp(ctypes.byref(hdrs[i]), ctypes.byref(fr[i]), ctypes.byref(so), pcm[i].ctypes.data_as(P16), 1,
                                                   want.ctypes.data_as(P16), 1)
                    else:
                        po = cap.PsState.from_buffer_copy(bytes(ps[i]))
                        rc = orc.lib.xo_sbr_dec_hq(ctypes.byref(hdrs[i]), ctypes.byref(fr[i]), ctypes.byref(so), ctypes.byref(pfs[i]),
                                                   ctypes.byref(po), pcm[i].ctypes.data_as(P16), 1, want.ctypes.data_as(P16), 2)
                    frames += 1
                    gs = cap.State.from_buffer_copy(stb[i].tobytes())
                    ok = status[i] == rc
                    if ok and rc == 0:
                        ok = np.array_equal(out[per * i:per * (i + 1)], want) and not cap.diff_state(gs, so)
                        if ok and mode == "hq":
                            ok = not cap.diff_state(cap.PsState.from_buffer_copy(psb[i].tobytes()), po)
                    if not ok:
                        bad += 1
                        print("MISMATCH seed", seed, mode, "step", step, "stream", i, "rc", rc, int(status[i]))
                    if rc == 0:
                        st[i] = so
                        if mode == "hq":
                            ps[i] = po
                    else:
                        st[i] = gs
                        if mode == "hq":
                            ps[i] = cap.PsState.from_buffer_copy(psb[i].tobytes())
    print("frames", frames, "bad", bad)


if __name__ == "__main__":
    main()
