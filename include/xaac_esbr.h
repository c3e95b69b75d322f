/*
 * xaac_esbr.h -- boundary formats of the eSBR ("Path A", the reference's default -esbr:1) SBR tool for HE-AAC streams:
 * what ixheaacd_sbr_dec's Path A branch (decoder/ixheaacd_sbr_dec.c:816-1009) reads beyond xaac_sbr_header /
 * xaac_sbr_frame (xaac_sbr.h), and the per-channel state it keeps between frames.
 * Scope: 2:1 SBR of AAC-LC cores (usac_flag = 0), with or without parametric stereo, and of USAC channels (usac_flag = 1,
 * stereoConfigIndex 0: the reference's USAC front end hands its core samples and SBR side info to the same seam), LPP or
 * harmonic patching (the QMF transposer of xaac_hbe.h), pre-flattening of LPP patches, inter-TES: no MPS.
 */
#ifndef XAAC_ESBR_H
#define XAAC_ESBR_H

#include <stdint.h>

#include "xaac_amd.h"
#include "xaac_sbr.h"
#include "xaac_hbe.h"
#include "xaac_pvc.h"

#define XAAC_ESBR_HIST_ROWS 40 /* op_delay 6 + SBR_HF_ADJ_OFFSET 2 + codec_x_delay 32 rows of qmf_buf_real/_imag kept */
#define XAAC_ESBR_OUT_HIST_ROWS 8 /* op_delay 6 + SBR_HF_ADJ_OFFSET 2 rows of sbr_qmf_out_real/_imag kept */
#define XAAC_ESBR_ROWS (XAAC_ESBR_HIST_ROWS + 32)
#define XAAC_ESBR_OUT_HIST_ROWS_4_1 14 /* 4:1 SBR: op_delay 12 (sbr_dec.c:719) + SBR_HF_ADJ_OFFSET 2 rows of both matrices kept: qmf_re /
                                          qmf_im rows 0..13; sbr_qmf_out's in out_re / out_im (rows 0..7) and ph_re / ph_im rows 0..5
                                          (rows 8..13) -- a 4:1 channel has no transposer in this library, whose rows those are */

/* bits of xaac_esbr_side::harmonic_sbr.  XAAC_ESBR_USAC: header usac_flag -- the clearing of the history rows above the old
   cross-over band is an AAC-only step (sbr_dec.c:868-874).  XAAC_ESBR_NO_X_DELAY: codec_x_delay = 0 (sbr_dec.c:819-826: a USAC
   channel without a harmonic transposer): the frame's 32 analysis rows are rows 8..39 of the QMF buffer, not rows 40..71 --
   the tools work six slots behind the analysis bank instead of thirty-eight, the state keeps eight history rows (rows 8..39
   of qmf_re / qmf_im are zero between frames, as in the reference's buffer).  XAAC_ESBR_SKIP_ADJUST: the frame's sbr_mode is
   not ORIG_SBR (UNKNOWN_SBR in a USAC channel's first frames): the HF generator runs, the envelope adjuster only does its
   reset and its end-of-frame bookkeeping (esbr_envcal.c:646: every envelope's work is inside `if (sbr_mode == ORIG_SBR)`). */
enum { XAAC_ESBR_HARMONIC = 1, XAAC_ESBR_PRE_FLATTEN = 2, XAAC_ESBR_USAC = 4, XAAC_ESBR_NO_X_DELAY = 8, XAAC_ESBR_SKIP_ADJUST = 16,
       XAAC_ESBR_OVERSAMPLING = 32 /* frame: over_sampling_flag (read by the DFT transposer only, sbr_dec.c:884) */ };

/* Per-frame side info: ia_sbr_header_data_struct / ia_freq_band_data_struct / ia_sbr_frame_info_data_struct members
 * (decoder/ixheaacd_env_extr_part.h:33-100, ixheaacd_env_extr.h:54-120) the float path reads and the fixed path does not. */
typedef struct xaac_esbr_side {
  int32_t out_sampling_freq;                          /* header: out_sampling_freq */
  int16_t limiter_bands;                              /* header: limiter_bands */
  int16_t num_mf_bands;                               /* freq band data: num_mf_bands */
  int16_t f_master_tbl[XAAC_SBR_MAX_FREQ_COEFFS + 1]; /* freq band data: f_master_tbl */
  int16_t qmf_sb_prev;                                /* freq band data: qmf_sb_prev (sbr_dec.c:314) */
  int16_t reset_flag;                                 /* frame: reset_flag */
  int16_t harmonic_sbr;                               /* what the frame's ENHSBR payload asks of the HF generator (env_extr.c:595-714):
                                                         XAAC_ESBR_HARMONIC: sbr_patching_mode == 0, it takes the harmonic
                                                         transposer's output, not LPP patches; XAAC_ESBR_PRE_FLATTEN: header
                                                         pre_proc_flag, LPP patches are pre-flattened (sbrdec_lpfuncs.c:928) */
  int32_t sbr_invf_mode_prev[XAAC_SBR_MAX_NOISE_VALUES]; /* frame: sbr_invf_mode_prev (set by the parser, env_extr.c:834) */
  int32_t inter_temp_shape_mode[XAAC_SBR_MAX_ENVELOPES]; /* frame: inter_temp_shape_mode (0 without inter-TES) */
  float flt_env_sf_arr[XAAC_SBR_MAX_ENV_VALUES];      /* frame: flt_env_sf_arr */
  float flt_noise_floor[XAAC_SBR_MAX_NOISE_VALUES];   /* frame: flt_noise_floor */
  int32_t pitch_in_bins;                              /* frame: pitch_in_bins (0 unless the ENHSBR payload carries one) */
} xaac_esbr_side;

/* Per-channel persistent state of the Path A branch. */
typedef struct xaac_esbr_state {
  xaac_esbr_ana_state ana;                            /* str_codec_qmf_bank (32-bit rings) */
  xaac_esbr_syn_state syn;                            /* str_synthesis_qmf_bank */
  float qmf_re[XAAC_ESBR_HIST_ROWS][64], qmf_im[XAAC_ESBR_HIST_ROWS][64];         /* qmf_buf_real / _imag rows 0..39 */
  float out_re[XAAC_ESBR_OUT_HIST_ROWS][64], out_im[XAAC_ESBR_OUT_HIST_ROWS][64]; /* sbr_qmf_out_real / _imag rows 0..7 */
  float bw_array_prev[XAAC_SBR_MAX_PATCHES];          /* frame_data: bw_array_prev */
  float e_gain[5][64], noise_buf[5][64];              /* frame_data: e_gain, noise_buf */
  int32_t lim_table[4][13], gate_mode[4];             /* frame_data: lim_table, gate_mode (remade at a reset frame) */
  int32_t harm_index, phase_index;
  int32_t esbr_start_up;                              /* header: esbr_start_up; 1 for a new stream */
  int32_t env_short_flag_prev;
  int32_t patch_start_subband[XAAC_SBR_MAX_PATCHES + 1], num_patches; /* frame_data: patch_param */
  int8_t harm_flag_prev[64];
  int32_t prev_sbr_patching_mode;                     /* frame_data: prev_sbr_patching_mode (0 for a new stream) */
  float ph_re[XAAC_ESBR_OUT_HIST_ROWS][64], ph_im[XAAC_ESBR_OUT_HIST_ROWS][64]; /* ph_vocod_qmf_real / _imag rows 0..7 after
                                                         the frame's shift (sbr_dec.c:859-868): the transposer's last rows */
} xaac_esbr_state;

/* ---- PVC frames of USAC channels (sbr_mode == PVC_SBR): what ixheaacd_sbr_dec (sbr_dec.c:931-953: energies of the low band,
 * ixheaacd_pvc_process) and the PVC branch of ixheaacd_sbr_env_calc (esbr_envcal.c:194-607) read and keep beyond the structs
 * above.  A batch that hands both in runs PVC frames on the device and keeps the bookkeeping ORIG_SBR frames leave for a
 * PVC frame that may follow (esbr_envcal.c:861-899, sbr_dec.c:947-953). */
#define XAAC_ESBR_PVC_COLS 48 /* MAX_FREQ_COEFFS_SBR: time slots a qmapped_pvc row holds */
enum { XAAC_ESBR_SBR_UNKNOWN = 0, XAAC_ESBR_SBR_ORIG = 1, XAAC_ESBR_SBR_PVC = 2 }; /* SBR_TYPE_ID, ixheaacd_sbrdecoder.h:66 */

typedef struct xaac_esbr_pvc_side {
  int16_t sbr_mode;                                   /* frame: sbr_mode (XAAC_ESBR_SBR_*) */
  int16_t sine_position;                              /* frame: sine_position (env_extr.c:288-294; 31 = ESC_SIN_POS) */
  int16_t sin_start_for_cur_top, sin_len_for_cur_top; /* frame: the sinusoids the frame before started beyond its own end */
  int16_t border_vec[XAAC_SBR_MAX_ENVELOPES + 1];     /* frame: str_pvc_frame_info.border_vec (PVC time slots) */
  int16_t freq_res[XAAC_SBR_MAX_ENVELOPES];           /* frame: str_pvc_frame_info.freq_res */
  int16_t pad_;
  xaac_pvc_frame pvc;                                 /* the PVC decoder's frame (xaac_pvc.h); read for sbr_mode PVC only */
} xaac_esbr_pvc_side;

typedef struct xaac_esbr_pvc_state {
  xaac_pvc_state pvc;                                 /* ia_pvc_data_struct: the decoder's history */
  float qmapped[64][XAAC_ESBR_PVC_COLS];              /* frame_data: qmapped_pvc */
  float prev_noise_level[XAAC_SBR_MAX_NOISE_VALUES];  /* frame_data: prev_noise_level */
  int8_t harm_flag_varlen_prev[64], harm_flag_varlen[64];
  int16_t prev_freq_res[2];                           /* frame_data: str_frame_info_prev.freq_res[0..1] (var_len_id_prev indexes it) */
  int16_t var_len_id_prev;
  int16_t prev_sbr_mode;                              /* frame_data: prev_sbr_mode */
  int32_t esbr_start_up_pvc;                          /* header: esbr_start_up_pvc; 1 for a new stream */
} xaac_esbr_pvc_state;

/* Per-stream persistent state of the float parametric-stereo tool (ia_ps_dec_struct's float members,
 * decoder/ixheaacd_ps_dec.h:162-237, 20-band configuration) + the right channel's synthesis bank. */
typedef struct xaac_esbr_ps_state {
  float hyb_hist_re[3][12], hyb_hist_im[3][12];       /* hyb_qmf_buf_re_20 / _im_20: 12 slots of QMF bands 0..2 */
  float qmf_delay_re[14][64], qmf_delay_im[14][64];   /* qmf_delay_buf_re / _im */
  float sub_delay_re[2][12], sub_delay_im[2][12];     /* sub_qmf_delay_buf_re / _im (rows 0..1, 12 hybrid sub-bands) */
  float ser_qmf_re[3][5][64], ser_qmf_im[3][5][64];   /* ser_qmf_delay_buf_re / _im */
  float ser_sub_re[3][5][12], ser_sub_im[3][5][12];   /* ser_sub_qmf_dealy_buf_re / _im */
  float h_prev[8][20];                                /* h11_re h12_re h21_re h22_re h11_im h12_im h21_im h22_im _prev;
                                                         h11_re / h12_re start at 1.0 (ps_dec_flt.c:349-352) */
  float peak_decay_fast[20], prev_nrg[20], prev_peak_diff[20];
  int32_t delay_buf_idx, delay_buf_idx_ser[3];
  int32_t delay_qmf_idx[64];                          /* delay_qmf_delay_buf_idx */
  xaac_esbr_syn_state syn_r;                          /* pstr_sbr_channel[1]: str_synthesis_qmf_bank */
} xaac_esbr_ps_state;

typedef struct xaac_esbr_sbr_batch {
  int32_t n_ch;
  const float *core;               /* [n_ch][1024] time_sample_buf in */
  const xaac_sbr_header *header;   /* [n_ch] */
  const xaac_sbr_frame *frame;     /* [n_ch] (apply_processing, grid, invf modes, harmonics) */
  const xaac_esbr_side *side;      /* [n_ch] */
  xaac_esbr_state *state;          /* [n_ch] in/out */
  float *out;                      /* [n_ch][2048] time_sample_buf out (left channel with PS) */
  const xaac_ps_frame *ps_frame;   /* [n_ch], or NULL together with ps_state / out_r: no parametric stereo.  With PS, a stream
                                      whose header.channel_mode is not PS_STEREO (3) has no right channel in this frame: its
                                      out_r row and its right bank's state are left alone, as the reference leaves that bank
                                      alone until PS starts (the caller doubles the left samples, api.c:3639-3660) */
  xaac_esbr_ps_state *ps_state;    /* [n_ch] in/out */
  float *out_r;                    /* [n_ch][2048] right channel (ps_dec->time_sample_buf[1]) */
  int32_t *status;                 /* optional [n_ch]: 0, or -1 where the reference returns an error */
  void *workspace;                 /* device scratch >= xaac_esbr_workspace_bytes(n_ch) */
  uint64_t workspace_bytes;
  xaac_hbe_state *hbe_state;        /* [n_ch] in/out, or NULL: the QMF harmonic transposer (xaac_hbe.h) of every channel,
                                       run on each processed frame as the reference does for non-USAC streams
                                       (sbr_dec.c:882-909).  Without it a frame with harmonic_sbr set is refused. */
  int32_t hbe_max_synth_size;       /* optional hint: 4 or 8 = no transposer of the batch has a larger bank (xaac_hbe_state::
                                       synth_size, known to the host from xaac_hbe_state_reinit): the banks kernel then takes less
                                       LDS per channel and more channels run per CU; a channel with a larger bank is refused
                                       (status -1).  0: any size. */
  const xaac_esbr_pvc_side *pvc_side; /* [n_ch], or NULL together with pvc_state: no channel of the batch has PVC frames (a frame
                                       whose side says sbr_mode PVC is then refused by its XAAC_ESBR_* flags' owner, the host) */
  xaac_esbr_pvc_state *pvc_state;   /* [n_ch] in/out */
  int32_t sbr_ratio;                /* XAAC_ESBR_RATIO_*: the SBR ratio of every channel of the batch (header: sbr_ratio_idx).
                                       2:1 (0): as described above.  8:3: the first 768 floats of a core row through the 24-channel
                                       analysis bank (sbr_dec.c:218), everything behind it as for 2:1.  4:1: 1024 core samples through
                                       the 16-channel bank into 64 slots, four to an envelope time slot (is_usf_4: sbrdec_lpfuncs.c:1036,
                                       esbr_envcal.c:152), out rows of 4096 floats, workspace of xaac_esbr_workspace_bytes_ratio;
                                       USAC channels without a transposer only (side flags XAAC_ESBR_USAC | _NO_X_DELAY; ps_frame and
                                       hbe_state NULL) -- a 4:1 channel with harmonic SBR stays the caller's */
  int32_t down_sample;              /* 1: the down-sampled synthesis bank(s) (32 channels; the reference's -dsample:1, or an output rate above
                                       48 kHz, sbrdec_initfuncs.c:622): out / out_r rows hold half the samples (1024; 2048 at 4:1), at the
                                       same row pitch */
  /* -esbr_hq:1: the DFT harmonic transposer (xaac_hbe.h: xaac_hbe_dft_state) in the place of the QMF one -- hand in these
     INSTEAD of hbe_state (all NULL otherwise).  ixheaacd_dft_hbe_apply runs on each processed frame (sbr_dec.c:880-892), with the
     frame's pitch_in_bins and XAAC_ESBR_OVERSAMPLING flag; the limiter bands take hbe_dft_state's x_over_qmf.  A channel whose
     sizes have no transform (hbe_dft_state.last_status -1) is refused for a frame that asks for harmonic patching. */
  xaac_hbe_dft_state *hbe_dft_state;       /* [n_ch] in/out */
  const xaac_hbe_dft_cfg *hbe_dft_cfg_tab; /* [n_cfg] */
  const float *hbe_dft_coef_re, *hbe_dft_coef_im; /* [n_cfg][64][128] */
  const int32_t *hbe_dft_cfg;              /* [n_ch] or NULL (all 0) */
} xaac_esbr_sbr_batch;
enum { XAAC_ESBR_RATIO_2_1 = 0, XAAC_ESBR_RATIO_8_3 = 1, XAAC_ESBR_RATIO_4_1 = 2 };

/* The hand-offs on either side of the branch in ixheaacd_dec_execute: the core decoder's PCM16 (after the 32 -> 16 bit
 * conversion of xaac_imdct_process_batch's XAAC_PCM_SBR mode, channels of an element interleaved) as floats, one plane per
 * channel (decoder/ixheaacd_api.c:3385-3432); and the branch's float output as 16-bit PCM, saturated to [-32768, 32767] and
 * truncated towards zero, two channels interleaved (ixheaacd_samples_sat, decoder/ixheaacd_decode_main.c:82-107; a mono
 * channel that the API then duplicates, api.c:3639-3660: left == right). */
typedef struct xaac_esbr_core_in_batch {
  int32_t n_ch;                    /* channel-frames in total, a multiple of ch_fac */
  int32_t ch_fac;                  /* 1 or 2: channels interleaved in pcm */
  const int16_t *pcm;              /* [n_ch / ch_fac][1024][ch_fac] */
  float *core;                     /* [n_ch][1024] */
} xaac_esbr_core_in_batch;

typedef struct xaac_esbr_pcm_out_batch {
  int32_t n;                       /* streams */
  int32_t stride;                  /* floats between consecutive streams' planes in left / right (>= 2048) */
  const float *left, *right;       /* stream i's planes: left + i * stride, right + i * stride; 2048 samples each */
  int16_t *pcm;                    /* [n][2048][2] */
} xaac_esbr_pcm_out_batch;

#ifdef __cplusplus
extern "C" {
#endif
XAAC_API int32_t xaac_esbr_core_from_pcm16_batch(xaac_ctx *ctx, const xaac_esbr_core_in_batch *batch);
XAAC_API int32_t xaac_esbr_pcm16_from_float_batch(xaac_ctx *ctx, const xaac_esbr_pcm_out_batch *batch);
/* One frame of every channel through the Path A branch of ixheaacd_sbr_dec (mono / stereo channels; with ps_* set:
 * HE-AACv2 streams, ixheaacd_esbr_apply_ps ps_dec_flt.c:389 between regrouping and the two synthesis banks):
 * history shift (sbr_dec.c:835-857), ixheaacd_esbr_analysis_filt_block, ixheaacd_generate_hf (sbrdec_lpfuncs.c:981),
 * ixheaacd_sbr_env_calc (esbr_envcal.c:71), ixheaacd_esbr_synthesis_regrp + the synthesis bank (sbr_dec.c:297 / :447). */
XAAC_API uint64_t xaac_esbr_workspace_bytes(int32_t n_ch);
XAAC_API uint64_t xaac_esbr_workspace_bytes_ratio(int32_t n_ch, int32_t sbr_ratio); /* the larger scratch of a 4:1 batch */
XAAC_API int32_t xaac_esbr_sbr_process_batch(xaac_ctx *ctx, const xaac_esbr_sbr_batch *batch);
#ifdef __cplusplus
}
#endif
#endif /* XAAC_ESBR_H */
