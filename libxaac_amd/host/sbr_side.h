/*
 * sbr_side.h -- host-side SBR / PS side-info decoder: the SBR extension payload of one AAC-LC frame -> the header tables
 * (xaac_sbr_header), per-channel frame data after delta decoding and dequantisation (xaac_sbr_frame) and the PS frame
 * (xaac_ps_frame) that the GPU entry points xaac_sbr_lp_process_batch / xaac_sbr_hq_process_batch take.
 * Restates, for AAC-LC + SBR (+ PS) streams decoded without the eSBR tools (the reference's -esbr:0 path: enh_sbr = 0,
 * no USAC, no ELD / LD, no error concealment frames), the frame-level part of ixheaacd_applysbr
 * (decoder/ixheaacd_sbrdecoder.c:313-760) and what it calls: header / grid / envelope / noise / sine / extension parsing
 * (ixheaacd_env_extr.c), delta decoding, limiting, coupling and dequantisation (ixheaacd_env_dec.c), the frequency band
 * tables (ixheaacd_freq_sca.c), patch and limiter tables (ixheaacd_sbrdec_lpfuncs.c:62-440), PS payload and index decoding
 * (ixheaacd_sbrdec_lpfuncs.c:561-735, ixheaacd_ps_bitdec.c:79-282).  CPU code.
 */
#ifndef XAAC_HOST_SBR_SIDE_H
#define XAAC_HOST_SBR_SIDE_H

#include <stdint.h>

#include "../../include/xaac_esbr.h"
#include "../../include/xaac_sbr.h"
#include "bits.h"

enum { XS_NOT_INITIALIZED = 0, XS_UPSAMPLING = 1, XS_ACTIVE = 2 }; /* sync_state (ixheaacd_sbr_const.h) */
enum { XS_SBR_MONO = 1, XS_SBR_STEREO = 2, XS_PS_STEREO = 3 };
enum { XS_COUPLING_OFF = 0, XS_COUPLING_LEVEL = 1, XS_COUPLING_BAL = 2 };

struct XsFrameInfo { /* ia_frame_info_struct */
  int16_t frame_class, num_env, transient_env, num_noise_env;
  int16_t border_vec[XAAC_SBR_MAX_ENVELOPES + 1], freq_res[XAAC_SBR_MAX_ENVELOPES];
  int16_t noise_border_vec[XAAC_SBR_MAX_NOISE_ENVELOPES + 1];
};

struct XsFrameData { /* the members of ia_sbr_frame_info_data_struct this path uses; never cleared between frames */
  XsFrameInfo fi;
  int16_t dir_env[XAAC_SBR_MAX_ENVELOPES], dir_noise[XAAC_SBR_MAX_NOISE_ENVELOPES];
  int32_t invf_mode[XAAC_SBR_MAX_NOISE_VALUES];
  int32_t coupling_mode, max_qmf_subband_aac;
  int16_t amp_res, num_env_sfac;
  uint8_t add_harmonics[XAAC_SBR_MAX_FREQ_COEFFS];
  int16_t env_sf[XAAC_SBR_MAX_ENV_VALUES];
  int16_t noise_floor[XAAC_SBR_MAX_NOISE_VALUES];
  /* the eSBR interpretation (-esbr:1, enh_sbr): float copies the float tools read, and what the ENHSBR extension carries */
  float flt_env_sf[XAAC_SBR_MAX_ENV_VALUES], flt_noise_floor[XAAC_SBR_MAX_NOISE_VALUES];
  float prev_noise_level_flt[XAAC_SBR_MAX_NOISE_COEFFS]; /* ia_sbr_frame_info_data_struct::prev_noise_level */
  int32_t invf_mode_prev[XAAC_SBR_MAX_NOISE_VALUES];
  int32_t patching_mode, over_sampling, pitch_in_bins, reset_flag, reset_flag_frame;
};

struct XsPrevData { /* ia_sbr_prev_frame_data_struct */
  int16_t sfb_nrg_prev[XAAC_SBR_MAX_FREQ_COEFFS], prev_noise_level[XAAC_SBR_MAX_NOISE_COEFFS];
  int32_t invf_mode[XAAC_SBR_MAX_NOISE_VALUES];
  int32_t end_position, coupling_mode, amp_res, max_qmf_subband_aac;
};

struct XsHeader { /* ia_sbr_header_data_struct + ia_freq_band_data_struct + ia_transposer_settings_struct */
  int sync_state, err_flag, err_flag_prev;
  int channel_mode, amp_res, start_freq, stop_freq, xover_band, freq_scale, alter_scale, noise_bands;
  int limiter_bands, limiter_gains, interpol_freq, smoothing_mode;
  int out_sampling_freq;
  int pre_flatten;         /* pre_proc_flag: the last ENHSBR element asked for pre-flattened LPP patches (env_extr.c:602) */
  int16_t num_sf_bands[2], num_nf_bands, num_mf_bands, sub_band_start, sub_band_end, num_lf_bands, num_if_bands;
  int16_t f_master[XAAC_SBR_MAX_FREQ_COEFFS + 1];
  int16_t tbl_lim[XAAC_SBR_MAX_LIMITERS + 1], tbl_lo[XAAC_SBR_MAX_FREQ_COEFFS / 2 + 1], tbl_hi[XAAC_SBR_MAX_FREQ_COEFFS + 1];
  int16_t tbl_noise[XAAC_SBR_MAX_NOISE_COEFFS + 1];
  int16_t num_patches, start_patch, stop_patch;
  int16_t bw_borders[XAAC_SBR_MAX_NOISE_VALUES];
  xaac_sbr_patch patch[XAAC_SBR_MAX_PATCHES];
};

struct XsPs { /* the members of ia_ps_dec_struct the payload decoder keeps */
  int enable_iid, enable_icc, enable_ext, iid_mode, icc_mode, iid_quant, freq_res_ipd, frame_class, num_env, data_present;
  int16_t border_position[XAAC_PS_MAX_ENV + 2];
  uint8_t iid_dt[XAAC_PS_MAX_ENV], icc_dt[XAAC_PS_MAX_ENV];
  int16_t iid_par[XAAC_PS_MAX_ENV + 2][XAAC_PS_BANDS_FINE], icc_par[XAAC_PS_MAX_ENV + 2][XAAC_PS_BANDS_FINE];
  int16_t iid_prev[XAAC_PS_BANDS_FINE], icc_prev[XAAC_PS_BANDS_FINE];
};

struct XsDecoder { /* one stream */
  int core_channels, ps_enable;
  int enh;                 /* the reference's default -esbr:1: payloads run one frame late, ENHSBR extension, float dequantisation */
  int qmf_sb_prev;         /* pstr_freq_band_data->qmf_sb_prev (sbrdec_initfuncs.c:649, sbrdecoder.c:892-986) */
  int qmf_sb_prev_frame;   /* ... as the frame decoded last found it */
  int reset_pitch;         /* at a frame with res.reset: pitch_in_bins the reset's transposer runs take */
  int prev_bytes, prev_ext_type;
  uint8_t prev_payload[272];
  XsHeader hdr;
  XsFrameData fd[2];
  XsPrevData prev[2];
  XsPs ps;
};

/* what one frame asks of the caller, beside the side info itself */
struct XsFrameResult {
  int apply;          /* sync_state == SBR_ACTIVE: the frames carry apply_processing = 1 */
  int reset;          /* ixheaacd_sbr_dec_reset ran (sbrdecoder.c:103): before this frame's GPU call the caller sets, in the
                         device state of channels 0 .. reset_channels-1, ph_index = 0, filt_buf_noise_e = 0, start_up = 1,
                         bw_array_prev = 0, syn_lsb = codec_usb = sub_band_start, syn_usb = sub_band_end */
  int reset_channels;
  int upsampling;     /* ixheaacd_prepare_upsamp ran (:254): syn_lsb = codec_usb = 32, syn_usb = 64 in those channels */
  int stereo;         /* the payload was a channel pair element's */
  int ps;             /* channel_mode == PS_STEREO: ps_frame is valid, output has two channels */
  int ps_start;       /* first PS frame after mono frames: xaac_sbr_state_handover with XAAC_HANDOVER_PS_START (:762-775) */
  int frame_ok;       /* frame_status after parsing */
};

/* a new stream: core sampling rate (the SBR range runs at twice that), core channels, PS allowed */
void xs_init(XsDecoder *d, int core_sampling_rate, int core_channels, int ps_enable, int enh = 0);

/* enh streams: the members of xaac_esbr_side of channel c for the frame xs_decode_frame decoded last */
void xs_export_esbr_side(const XsDecoder *d, int c, xaac_esbr_side *o);

/* the QMF transposer's first synthesis band for SBR start band b (0 .. 32): ixheaac_start_subband2kL_tbl */
int xs_hbe_k_start(int start_band);

/* One frame.  payload / bytes / ext_type: XhElement::sbr etc. (bytes = 0: no SBR payload in this frame).
   Fills header, frame[0 .. 1], ps_frame and res.  Returns 0, or a negative value where the reference would have
   returned a fatal error from ixheaacd_applysbr. */
int xs_decode_frame(XsDecoder *d, const uint8_t *payload, int bytes, int ext_type, xaac_sbr_header *header,
                    xaac_sbr_frame frame[2], xaac_ps_frame *ps_frame, XsFrameResult *res);

/* after the GPU has run the frame: what ixheaacd_sbr_dec leaves in the previous-frame data (sbr_dec.c:1210-1218) */
void xs_frame_done(XsDecoder *d, const XsFrameResult *res);

#endif /* XAAC_HOST_SBR_SIDE_H */
