"""ctypes mirror of include/xaac_hbe.h (checked against the header's layout by tests/test_abi.py)."""
import ctypes

F, I32 = ctypes.c_float, ctypes.c_int32
NO_BINS = 32


class HbeState(ctypes.Structure):
    _fields_ = [("input_buf", F * (1024 + 64)), ("synth_buf", F * 1280), ("analy_buf", F * 640),
                ("qmf_in_buf", (F * 128) * NO_BINS), ("qmf_out_buf", (F * 128) * (2 * NO_BINS)),
                ("synth_size", I32), ("k_start", I32), ("start_band", I32), ("end_band", I32),
                ("x_over_qmf", I32 * 6), ("max_stretch", I32), ("fft_ready", I32)]


class HbeDftState(ctypes.Structure):
    _fields_ = [("analy_buf", F * 640), ("analy_size", I32), ("a_start", I32)]


K_START = [0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 4, 4, 4, 4, 4, 6, 6, 6, 8, 8, 8, 8, 8, 10, 10, 10, 12, 12, 12, 12, 12, 12, 12]


def new_state(start_band, end_band=None):
    """a fresh stream's state with the bank parameters of hbe_trans.c:107-112 for a start band"""
    st = HbeState()
    st.start_band = start_band
    st.end_band = min(64, 2 * start_band + 8) if end_band is None else end_band
    st.synth_size = 4 * ((start_band + 4) // 8 + 1)
    st.k_start = K_START[start_band]
    return st


def state_from_tables(lo, hi, prev_max_stretch=0):
    """what ixheaacd_qmf_hbe_data_reinit (hbe_trans.c:102-222) derives from an SBR header's low / high resolution
    frequency-band tables for 2:1 SBR of a 1024-sample core, on a fresh transposer (tests/test_hbe_oracle_vs_reference.py
    checks it against the reference's function)"""
    lo, hi = [int(v) for v in lo], [int(v) for v in hi]
    n_lo, n_hi = len(lo) - 1, len(hi) - 1
    st = HbeState()
    st.start_band, st.end_band = lo[0], lo[n_lo]
    st.synth_size = 4 * ((st.start_band + 4) // 8 + 1)
    st.k_start = K_START[st.start_band]
    st.max_stretch = prev_max_stretch
    sfb = 0
    for patch in range(1, 5):
        while sfb <= n_lo and lo[sfb] <= patch * st.start_band:
            sfb += 1
        if sfb <= n_lo:
            if patch * st.start_band - lo[sfb - 1] <= 3:
                st.x_over_qmf[patch - 1] = lo[sfb - 1]
            else:
                s2 = 0
                while s2 <= n_hi and hi[s2] <= patch * st.start_band:
                    s2 += 1
                st.x_over_qmf[patch - 1] = hi[s2 - 1]
        else:
            st.x_over_qmf[patch - 1] = st.end_band
            st.max_stretch = min(patch, 4)
            break
    return st


class HbeDftFullState(ctypes.Structure):   # xaac_hbe_dft_state
    _fields_ = [("input_buf", F * 1024), ("output_buf", F * 3072), ("synth_buf", F * 1280), ("anal", HbeDftState),
                ("synth_size", I32), ("k_start", I32), ("start_band", I32), ("end_band", I32), ("max_stretch", I32),
                ("x_over_qmf", I32 * 6), ("last_status", I32)]


class HbeDftCfg(ctypes.Structure):         # xaac_hbe_dft_cfg
    _fields_ = [("anal_window", F * 512), ("synth_window", F * 768), ("fd_win", ((F * 772) * 2) * 3)]
