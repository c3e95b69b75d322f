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


class PsFrame(ctypes.Structure):
    _fields_ = [("iid_quant", I16), ("freq_res_ipd", I16), ("border_position", I16 * 7), ("num_env", I16),
                ("iid_par_table", (I16 * 34) * 7), ("icc_par_table", (I16 * 34) * 7)]


class PsState(ctypes.Structure):
    _fields_ = [("ser", ((I16 * 64) * 3) * 5), ("ap", (I16 * 64) * 2), ("ld", (I16 * 24) * 14), ("sd", I16 * 64),
                ("sub", (I16 * 32) * 2), ("sub_ser", ((I16 * 32) * 3) * 5), ("idx_ser", I16 * 3), ("sample_ser", I16 * 3),
                ("idx", I16), ("idx_long", I16), ("peak_decay_diff", I32 * 20), ("energy_prev", I32 * 20),
                ("peak_decay_diff_prev", I32 * 20), ("hyb_buf", ((I32 * 12) * 2) * 3), ("h11_h12_vec", I16 * 48),
                ("h21_h22_vec", I16 * 48), ("H11_H12", I16 * 48), ("H21_H22", I16 * 48), ("delta_h11_h12", I16 * 48),
                ("delta_h21_h22", I16 * 48), ("delay_buffer_scale", I16), ("usb", I16), ("syn_ring_r", I16 * 1280),
                ("syn_drc_offset_r", I16), ("syn_phase_r", I16), ("syn_lsb_r", I16), ("syn_usb_r", I16),
                ("st_syn_scale_r", I16), ("lb_scale_r", I16), ("ov_lb_scale_r", I16), ("hb_scale_r", I16)]


def diff_state(a, b):
    """list of (field, detail) where two State structs differ"""
    out = []
    for n, _ in type(a)._fields_:
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
            rec = dict(call=int(meta[1]), low_pow=int(meta[2]), ch_fac=int(meta[3]), aot=int(meta[4]),
                       ps=int(meta[5]), ret=int(meta[6]), enh=int(meta[7]), header=hd, frame=fr, st0=st0,
                       st1=st1, pcm_in=pcm_in, pcm_out=pcm_out)
            if rec["ps"]:       # HE-AACv2 records carry the PS side info and state (oracle/ref_capture.c)
                rec["ps_frame"] = PsFrame.from_buffer_copy(f.read(ctypes.sizeof(PsFrame)))
                rec["ps0"] = PsState.from_buffer_copy(f.read(ctypes.sizeof(PsState)))
                rec["ps1"] = PsState.from_buffer_copy(f.read(ctypes.sizeof(PsState)))
            recs.append(rec)
            if limit and len(recs) >= limit:
                break
    return recs


# ---- AAC-ELD channels (low-delay SBR): include/xaac_amd.h xaac_sbr_eld_state
class EldAna(ctypes.Structure):
    _fields_ = [("ring", I16 * 320), ("wr", I16), ("f1", I16), ("f2", I16), ("fp", I16)]


class EldSyn(ctypes.Structure):
    _fields_ = [("ring", I16 * 1280), ("drc_offset", I16), ("phase", I16), ("fp", I16), ("sixty4", I16)]


class EldState(ctypes.Structure):
    _fields_ = [("ana", EldAna), ("syn", EldSyn), ("codec_usb", I16), ("syn_lsb", I16), ("syn_usb", I16), ("pad2_", I16)] + \
               State._fields_[State._fields_.index(("lpc_real", (I32 * 32) * 2)):]


def eld_state_from(st):
    """a new AAC-ELD channel's state: the banks as sbrdec_initfuncs.c:1122-1148 / :1181-1209 leave them, the rest (the
    members ixheaacd_sbr_dec's core works on) taken from a captured xaac_sbr_state image of the reference's structs"""
    e = EldState()
    e.ana.f2, e.syn.sixty4 = 32, 64
    e.codec_usb, e.syn_lsb, e.syn_usb = st.codec_usb, st.syn_lsb, st.syn_usb
    off = State.lpc_real.offset
    ctypes.memmove(ctypes.addressof(e) + EldState.lpc_real.offset, ctypes.addressof(st) + off, ctypes.sizeof(State) - off)
    return e
