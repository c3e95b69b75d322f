/*
 * xaac_sbr.h -- plain-C data formats that cross the boundary for the fixed-point SBR chain
 * (ixheaacd_sbr_dec "Path B", decoder/ixheaacd_sbr_dec.c:662): what the CPU-side bitstream
 * parser hands over per frame, and the per-channel persistent state that lives on the GPU.
 * Every struct flattens the reference structs it names (pointers become inline arrays /
 * offsets) so a reference decoder instance can be copied into it field by field; the
 * reference-side adapter that does exactly that is oracle/ref_capture.c.
 */
#ifndef XAAC_SBR_H
#define XAAC_SBR_H

#include <stdint.h>

/* The libraries are built with -fvisibility=hidden: the functions these headers declare are everything they export
   (tests/test_abi.py compares `nm -D` with the declarations). */
#ifndef XAAC_API
#define XAAC_API __attribute__((visibility("default")))
#endif

#define XAAC_SBR_MAX_ENVELOPES 8       /* MAX_ENVELOPES, decoder/ixheaacd_sbrdecsettings.h:50 */
#define XAAC_SBR_MAX_NOISE_ENVELOPES 2
#define XAAC_SBR_MAX_FREQ_COEFFS 56
#define XAAC_SBR_MAX_NOISE_COEFFS 5
#define XAAC_SBR_MAX_LIMITERS 12
#define XAAC_SBR_MAX_PATCHES 6
#define XAAC_SBR_MAX_ENV_VALUES (XAAC_SBR_MAX_ENVELOPES * XAAC_SBR_MAX_FREQ_COEFFS)
#define XAAC_SBR_MAX_NOISE_VALUES (XAAC_SBR_MAX_NOISE_ENVELOPES * XAAC_SBR_MAX_NOISE_COEFFS)

/* Header-derived tables; change only on an SBR header / reset.
 * = ia_sbr_header_data_struct scalars (decoder/ixheaacd_env_extr_part.h:51-100)
 * + ia_freq_band_data_struct (:33-49) + ia_transposer_settings_struct (decoder/ixheaacd_lpp_tran.h:42-56)
 * (the lsb/usb of the two QMF banks follow max_qmf_subband_aac frame by frame -- sbrdec_lpfuncs.c:470-471 --
 * and therefore live in xaac_sbr_state). */
typedef struct xaac_sbr_patch {
  int16_t src_start_band, src_end_band, guard_start_band, dst_start_band, dst_end_band, num_bands_in_patch;
} xaac_sbr_patch;

typedef struct xaac_sbr_header {
  int16_t num_time_slots, time_step;          /* 16, 2 */
  int16_t channel_mode;                       /* SBR_MONO 1 / SBR_STEREO 2 / PS_STEREO 3 */
  int16_t limiter_gains, interpol_freq, smoothing_mode;
  int16_t num_sf_bands[2];                    /* [LOW], [HIGH] */
  int16_t num_nf_bands, sub_band_start, sub_band_end, num_lf_bands, num_if_bands;
  int16_t freq_band_tbl_lim[XAAC_SBR_MAX_LIMITERS + 1];
  int16_t freq_band_tbl_lo[XAAC_SBR_MAX_FREQ_COEFFS / 2 + 1];
  int16_t freq_band_tbl_hi[XAAC_SBR_MAX_FREQ_COEFFS + 1];
  int16_t freq_band_tbl_noise[XAAC_SBR_MAX_NOISE_COEFFS + 1];
  int16_t num_columns, num_patches, start_patch, stop_patch;
  int16_t bw_borders[XAAC_SBR_MAX_NOISE_VALUES];
  xaac_sbr_patch patch[XAAC_SBR_MAX_PATCHES];
} xaac_sbr_header;

/* Per-channel, per-frame side info after dequantisation (ia_sbr_frame_info_data_struct,
 * decoder/ixheaacd_env_extr.h:54-75, with ia_frame_info_struct, env_extr_part.h:102-110). */
typedef struct xaac_sbr_frame {
  int16_t num_env, transient_env, num_noise_env, frame_class;
  int16_t border_vec[XAAC_SBR_MAX_ENVELOPES + 1];
  int16_t freq_res[XAAC_SBR_MAX_ENVELOPES];
  int16_t noise_border_vec[XAAC_SBR_MAX_NOISE_ENVELOPES + 1];
  int16_t amp_res;
  int16_t apply_processing;                   /* FLAG passed to ixheaacd_sbr_dec */
  int32_t coupling_mode;
  int32_t max_qmf_subband_aac;
  int32_t sbr_invf_mode[XAAC_SBR_MAX_NOISE_VALUES];
  uint8_t add_harmonics[XAAC_SBR_MAX_FREQ_COEFFS];
  int16_t int_env_sf_arr[XAAC_SBR_MAX_ENV_VALUES];    /* packed mantissa/exponent, env_extr.h:25-33 */
  int16_t int_noise_floor[XAAC_SBR_MAX_NOISE_VALUES];
} xaac_sbr_frame;

/* The members of xaac_sbr_state behind the filterbank rings and the overlap slots -- the part the
 * serial SBR core reads and writes.  Spelled as a macro so that the GPU core can keep a copy of
 * exactly this tail in LDS (libxaac_amd/csrc/sbr_core_kernel.hip) without a second field list. */
#define XAAC_SBR_STATE_TAIL_FIELDS                                                                         \
  int32_t lpc_real[2][32], lpc_imag[2][32]; /* str_hf_generator.lpc_filt_states_* */                       \
  int32_t bw_array_prev[XAAC_SBR_MAX_PATCHES];                                                             \
  int16_t lb_scale, st_lb_scale, ov_lb_scale, hb_scale, ov_hb_scale, st_syn_scale, ps_scale, pad0_;        \
  /* ia_sbr_prev_frame_data_struct members the path reads/writes */                                        \
  int32_t prev_invf_mode[XAAC_SBR_MAX_NOISE_VALUES];                                                       \
  int32_t prev_max_qmf_subband_aac;                                                                        \
  int32_t prev_coupling_mode;                                                                              \
  int16_t prev_end_position, prev_amp_res;                                                                 \
  /* ia_sbr_calc_env_struct */                                                                             \
  int16_t filt_buf_me[2 * XAAC_SBR_MAX_FREQ_COEFFS];                                                       \
  int16_t filt_buf_noise_m[XAAC_SBR_MAX_FREQ_COEFFS];                                                      \
  int32_t filt_buf_noise_e;                                                                                \
  int32_t start_up;                                                                                        \
  int16_t ph_index, tansient_env_prev, harm_index, pad1_;                                                  \
  int8_t harm_flags_prev[XAAC_SBR_MAX_FREQ_COEFFS];

/* Per-channel persistent state of ixheaacd_sbr_dec. */
typedef struct xaac_sbr_state {
  /* analysis / synthesis banks: same layouts as xaac_qmf_ana_state / xaac_qmf_syn_state */
  int16_t ana_ring[320], ana_wr, ana_phase;
  int16_t syn_ring[1280], syn_drc_offset, syn_phase;
  int16_t codec_usb;                          /* str_codec_qmf_bank.usb (lsb is 0) */
  int16_t syn_lsb, syn_usb;                   /* str_synthesis_qmf_bank.lsb / .usb */
  int16_t pad2_;
  int32_t overlap[6 * 64 * 2];                /* ptr_sbr_overlap_buf: 6 slots x 64 (LP) or x 128 (HQ) */
  XAAC_SBR_STATE_TAIL_FIELDS
} xaac_sbr_state;

/* ---- parametric stereo (HE-AACv2): ia_ps_dec_struct, decoder/ixheaacd_ps_dec.h:97-243 -------------------- */
#define XAAC_PS_MAX_ENV 5       /* MAXIM_NUM_OF_PS_ENVLOPS */
#define XAAC_PS_BANDS_FINE 34   /* NUM_BANDS_FINE */
#define XAAC_PS_GROUPS 22       /* NO_IID_GROUPS: 10 hybrid + 12 QMF groups */

/* Per-frame PS side info after ixheaacd_decode_ps_data (ixheaacd_ps_bitdec.c): the envelope borders and
 * the IID / ICC indices already mapped to the 20-band resolution the fixed-point path works in. */
typedef struct xaac_ps_frame {
  int16_t iid_quant;                                   /* fine (1) or coarse (0) IID quantiser */
  int16_t freq_res_ipd;                                /* = iid_mode 0..2 (ps_bitdec: freq_res_ipd); read by the float tool
                                                          of the eSBR path only (xaac_esbr.h), ignored by the fixed-point one */
  int16_t border_position[XAAC_PS_MAX_ENV + 2];
  int16_t num_env;                                     /* ps_dec->num_env after ixheaacd_decode_ps_data (1..5; border 0 is
                                                          0, border num_env is 32); read by the float tool only */
  int16_t iid_par_table[XAAC_PS_MAX_ENV + 2][XAAC_PS_BANDS_FINE];
  int16_t icc_par_table[XAAC_PS_MAX_ENV + 2][XAAC_PS_BANDS_FINE];
} xaac_ps_frame;

/* The PS tool's own part of xaac_ps_state, as a macro so that the GPU kernel can keep exactly these members in
 * LDS (libxaac_amd/csrc/sbr_ps_kernel.hip) without a second field list. */
#define XAAC_PS_STATE_HEAD_FIELDS                                                                           \
  int16_t ser[5][3][64];     /* delay_buf_qmf_ser_re_im: all-pass links of the QMF bands (re,im pairs) */   \
  int16_t ap[2][64];         /* delay_buf_qmf_ap_re_im */                                                   \
  int16_t ld[14][24];        /* delay_buf_qmf_ld_re_im: 14-slot delay, 12 bands */                          \
  int16_t sd[64];            /* delay_buf_qmf_sd_re_im: 1-slot delay (58 used; contiguous with ld) */       \
  int16_t sub[2][32];        /* delay_buf_qmf_sub_re_im: hybrid sub-bands */                                \
  int16_t sub_ser[5][3][32]; /* delay_buf_qmf_sub_ser_re_im (contiguous with sub) */                        \
  int16_t idx_ser[3], sample_ser[3];                                                                        \
  int16_t idx, idx_long;                                                                                    \
  int32_t peak_decay_diff[20], energy_prev[20], peak_decay_diff_prev[20]; /* contiguous, in this order */   \
  int32_t hyb_buf[3][2][12]; /* str_hybrid.ptr_qmf_buf_re/_im: 12-slot history of QMF bands 0..2 */         \
  int16_t h11_h12_vec[48], h21_h22_vec[48], H11_H12[48], H21_H22[48], delta_h11_h12[48], delta_h21_h22[48]; \
  int16_t delay_buffer_scale, usb;

/* Per-stream persistent PS state + the right channel's synthesis bank. */
typedef struct xaac_ps_state {
  XAAC_PS_STATE_HEAD_FIELDS
  /* right channel: str_synthesis_qmf_bank + scale factors of pstr_sbr_channel[1] */
  int16_t syn_ring_r[1280], syn_drc_offset_r, syn_phase_r;
  int16_t syn_lsb_r, syn_usb_r;
  int16_t st_syn_scale_r, lb_scale_r, ov_lb_scale_r, hb_scale_r;
} xaac_ps_state;

#endif /* XAAC_SBR_H */
