"""ctypes mirrors of include/xaac_esbr.h (test infrastructure)."""
import ctypes

I8, I16, I32, F32 = ctypes.c_int8, ctypes.c_int16, ctypes.c_int32, ctypes.c_float


class EsbrSide(ctypes.Structure):
    _fields_ = [("out_sampling_freq", I32), ("limiter_bands", I16), ("num_mf_bands", I16), ("f_master_tbl", I16 * 57),
                ("qmf_sb_prev", I16), ("reset_flag", I16), ("harmonic_sbr", I16), ("sbr_invf_mode_prev", I32 * 10),
                ("inter_temp_shape_mode", I32 * 8), ("flt_env_sf_arr", F32 * 448), ("flt_noise_floor", F32 * 10), ("pitch_in_bins", I32)]


class EsbrAna(ctypes.Structure):
    _fields_ = [("ring", I32 * 320), ("pos", I32), ("win_off", I32)]


class EsbrSyn(ctypes.Structure):
    _fields_ = [("ring", I32 * 1280), ("drc_offset", I32), ("filt_off", I32)]


class EsbrState(ctypes.Structure):
    _fields_ = [("ana", EsbrAna), ("syn", EsbrSyn), ("qmf_re", (F32 * 64) * 40), ("qmf_im", (F32 * 64) * 40),
                ("out_re", (F32 * 64) * 8), ("out_im", (F32 * 64) * 8), ("bw_array_prev", F32 * 6),
                ("e_gain", (F32 * 64) * 5), ("noise_buf", (F32 * 64) * 5), ("lim_table", (I32 * 13) * 4),
                ("gate_mode", I32 * 4), ("harm_index", I32), ("phase_index", I32), ("esbr_start_up", I32),
                ("env_short_flag_prev", I32), ("patch_start_subband", I32 * 7), ("num_patches", I32),
                ("harm_flag_prev", I8 * 64), ("prev_sbr_patching_mode", I32), ("ph_re", (F32 * 64) * 8),
                ("ph_im", (F32 * 64) * 8)]


def new_state():
    st = EsbrState()
    st.esbr_start_up = 1
    return st


class EsbrPsState(ctypes.Structure):
    _fields_ = [("hyb_hist_re", (F32 * 12) * 3), ("hyb_hist_im", (F32 * 12) * 3), ("qmf_delay_re", (F32 * 64) * 14),
                ("qmf_delay_im", (F32 * 64) * 14), ("sub_delay_re", (F32 * 12) * 2), ("sub_delay_im", (F32 * 12) * 2),
                ("ser_qmf_re", ((F32 * 64) * 5) * 3), ("ser_qmf_im", ((F32 * 64) * 5) * 3),
                ("ser_sub_re", ((F32 * 12) * 5) * 3), ("ser_sub_im", ((F32 * 12) * 5) * 3), ("h_prev", (F32 * 20) * 8),
                ("peak_decay_fast", F32 * 20), ("prev_nrg", F32 * 20), ("prev_peak_diff", F32 * 20),
                ("delay_buf_idx", I32), ("delay_buf_idx_ser", I32 * 3), ("delay_qmf_idx", I32 * 64), ("syn_r", EsbrSyn)]


def new_ps_state():
    st = EsbrPsState()
    for j in range(20):
        st.h_prev[0][j] = 1.0
        st.h_prev[1][j] = 1.0
    return st


# ---- PVC frames of USAC channels (include/xaac_esbr.h: xaac_esbr_pvc_side / xaac_esbr_pvc_state)
from pvc_structs import PvcFrame, PvcState  # noqa: E402


class EsbrPvcSide(ctypes.Structure):
    _fields_ = [("sbr_mode", I16), ("sine_position", I16), ("sin_start_for_cur_top", I16), ("sin_len_for_cur_top", I16),
                ("border_vec", I16 * 9), ("freq_res", I16 * 8), ("pad_", I16), ("pvc", PvcFrame)]


class EsbrPvcState(ctypes.Structure):
    _fields_ = [("pvc", PvcState), ("qmapped", (F32 * 48) * 64), ("prev_noise_level", F32 * 10), ("harm_flag_varlen_prev", I8 * 64),
                ("harm_flag_varlen", I8 * 64), ("prev_freq_res", I16 * 2), ("var_len_id_prev", I16), ("prev_sbr_mode", I16),
                ("esbr_start_up_pvc", I32)]
