"""Reader for the records written by oracle/ref_capture.c (formats of include/xaac_sbr.h).
Test infrastructure."""
import ctypes

import numpy as np

I16, I32, U8, I8 = ctypes.c_int16, ctypes.c_int32, ctypes.c_uint8, ctypes.c_int8


class Patch(ctypes.Structure):
    _fields_ = [(n, I16) for n in ("src_start_band", "src_end_band", "guard_start_band", "dst_start_band",
                                   "dst_end_band", "num_bands_in_patch")]


class Header(ctypes.Structure):
    _fields_ = [("num_time_slots", I16), ("time_step", I16), ("channel_mode", I16), ("limiter_gains", I16),
                ("interpol_freq", I16), ("smoothing_mode", I16), ("num_sf_bands", I16 * 2), ("num_nf_bands", I16),
                ("sub_band_start", I16), ("sub_band_end", I16), ("num_lf_bands", I16), ("num_if_bands", I16),
                ("freq_band_tbl_lim", I16 * 13), ("freq_band_tbl_lo", I16 * 29), ("freq_band_tbl_hi", I16 * 57),
                ("freq_band_tbl_noise", I16 * 6), ("num_columns", I16), ("num_patches", I16), ("start_patch", I16),
                ("stop_patch", I16), ("bw_borders", I16 * 10), ("patch", Patch * 6)]


class Frame(ctypes.Structure):
    _fields_ = [("num_env", I16), ("transient_env", I16), ("num_noise_env", I16), ("frame_class", I16),
                ("border_vec", I16 * 9), ("freq_res", I16 * 8), ("noise_border_vec", I16 * 3), ("amp_res", I16),
                ("apply_processing", I16), ("coupling_mode", I32), ("max_qmf_subband_aac", I32),
                ("sbr_invf_mode", I32 * 10), ("add_harmonics", U8 * 56), ("int_env_sf_arr", I16 * 448),
                ("int_noise_floor", I16 * 10)]


class State(ctypes.Structure):
    _fields_ = [("ana_ring", I16 * 320), ("ana_wr", I16), ("ana_phase", I16), ("syn_ring", I16 * 1280),
                ("syn_drc_offset", I16), ("syn_phase", I16), ("codec_usb", I16), ("syn_lsb", I16), ("syn_usb", I16), ("pad2_", I16),
                ("overlap", I32 * 768), ("lpc_real", (I32 * 32) * 2),
                ("lpc_imag", (I32 * 32) * 2), ("bw_array_prev", I32 * 6), ("lb_scale", I16), ("st_lb_scale", I16),
                ("ov_lb_scale", I16), ("hb_scale", I16), ("ov_hb_scale", I16), ("st_syn_scale", I16),
                ("ps_scale", I16), ("pad0_", I16), ("prev_invf_mode", I32 * 10), ("prev_max_qmf_subband_aac", I32),
                ("prev_coupling_mode", I32), ("prev_end_position", I16), ("prev_amp_res", I16),
                ("filt_buf_me", I16 * 112), ("filt_buf_noise_m", I16 * 56), ("filt_buf_noise_e", I32),
                ("start_up", I32), ("ph_index", I16), ("tansient_env_prev", I16), ("harm_index", I16), ("pad1_", I16),
                ("harm_flags_prev", I8 * 56)]


def diff_state(a, b):
    """list of (field, detail) where two State structs differ"""
    out = []
    for n, _ in State._fields_:
        va, vb = getattr(a, n), getattr(b, n)
        if hasattr(va, "__len__"):
            xa, xb = np.ctypeslib.as_array(va).ravel(), np.ctypeslib.as_array(vb).ravel()
            if not np.array_equal(xa, xb):
                out.append((n, int(np.sum(xa != xb)), np.nonzero(xa != xb)[0][:6].tolist()))
        elif va != vb:
            out.append((n, va, vb))
    return out


def read_records(path, limit=None):
    import gzip
    recs = []
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        while True:
            m = f.read(32)
            if len(m) < 32:
                break
            meta = np.frombuffer(m, np.int32)
            assert meta[0] == 0x58414331
            hd = Header.from_buffer_copy(f.read(ctypes.sizeof(Header)))
            fr = Frame.from_buffer_copy(f.read(ctypes.sizeof(Frame)))
            st0 = State.from_buffer_copy(f.read(ctypes.sizeof(State)))
            pcm_in = np.frombuffer(f.read(2048), np.int16).copy()
            st1 = State.from_buffer_copy(f.read(ctypes.sizeof(State)))
            pcm_out = np.frombuffer(f.read(8192), np.int16).reshape(2, 2048).copy()
            recs.append(dict(call=int(meta[1]), low_pow=int(meta[2]), ch_fac=int(meta[3]), aot=int(meta[4]),
                             ps=int(meta[5]), ret=int(meta[6]), enh=int(meta[7]), header=hd, frame=fr, st0=st0,
                             st1=st1, pcm_in=pcm_in, pcm_out=pcm_out))
            if limit and len(recs) >= limit:
                break
    return recs
