/*
 * xaac_hbe.h -- boundary formats of the QMF-domain harmonic transposer ("HBE") of the eSBR tool: the per-channel state
 * of ia_esbr_hbe_txposer_struct (decoder/ixheaacd_sbr_dec.h:30-100) that ixheaacd_qmf_hbe_apply
 * (decoder/ixheaacd_hbe_trans.c:224) keeps between frames, and batched entry points for its two polyphase banks
 * (decoder/ixheaacd_esbr_polyphase.c):
 *
 *   xaac_hbe_real_synth_batch  <-> ixheaacd_real_synth_filt     (esbr_polyphase.c:157-274)
 *   xaac_hbe_cplx_anal_batch   <-> ixheaacd_complex_anal_filt   (esbr_polyphase.c:48-155)
 *   xaac_hbe_apply_batch       <-> ixheaacd_qmf_hbe_apply       (hbe_trans.c:224-296)
 *
 * Scope: the QMF transposer (esbr_hq = 0) at 2:1 SBR of 1024-sample cores: no_bins = 32 QMF columns per frame, bank
 * sizes synth_size = 4, 8, 12, 16, 20 (hbe_trans.c:111-112).  The DFT transposer's bank (esbr_polyphase.c:276) is not
 * built.  All samples are FLOAT32; results are bit-identical to the reference's x86-64 build (float operations in the
 * reference's order, no contraction).
 */
#ifndef XAAC_HBE_H
#define XAAC_HBE_H

#include <stdint.h>

#include "xaac_amd.h"

#define XAAC_HBE_NO_BINS 32       /* QMF columns per frame (sbrdec_initfuncs.c:101) */
#define XAAC_HBE_OPER_WIN_LEN 13  /* HBE_OPER_WIN_LEN */

/* Per-channel state.  A new stream: all zero, then the four bank parameters as ixheaacd_qmf_hbe_data_reinit
   (hbe_trans.c:102-222) derives them from the SBR frequency tables. */
typedef struct xaac_hbe_state {
  float input_buf[1024 + 64];   /* ptr_input_buf: the core band's sub-sampled time signal (sbrdec_initfuncs.c:107-110) */
  float synth_buf[1280];        /* synth_buf: the real synthesis bank's 20 * synth_size delay line */
  float analy_buf[640];         /* analy_buf: the complex analysis bank's 20 * synth_size delay line */
  float qmf_in_buf[XAAC_HBE_NO_BINS][128];      /* qmf_in_buf rows: 64 (re, im) pairs */
  float qmf_out_buf[2 * XAAC_HBE_NO_BINS][128]; /* qmf_out_buf rows */
  int32_t synth_size;           /* 4 * ((start_band + 4) / 8 + 1) */
  int32_t k_start;              /* ixheaac_start_subband2kL_tbl[start_band] */
  int32_t start_band, end_band; /* freq_band_table[LOW][0], [LOW][num_sf_bands[LOW]] */
  int32_t x_over_qmf[6];        /* x_over_qmf */
  int32_t max_stretch;
  int32_t fft_ready;            /* the reference's FFT pointers are set: see xaac_hbe_apply_batch */
} xaac_hbe_state;

typedef struct xaac_hbe_synth_batch {
  int32_t n_ch;
  int32_t num_columns;          /* 32 */
  const float *qmf_re, *qmf_im; /* [n_ch][num_columns][64]: qmf_buf_real / _imag rows handed to ixheaacd_qmf_hbe_apply */
  xaac_hbe_state *state;        /* [n_ch] in/out: synth_buf; input_buf[(idx + 1) * synth_size ..] written per column */
  int32_t *status;              /* [n_ch] or NULL: 0, or -1 for bank parameters outside the tables */
} xaac_hbe_synth_batch;

typedef struct xaac_hbe_apply_batch_desc {
  int32_t n_ch;
  const float *qmf_re, *qmf_im; /* [n_ch][32][64]: the frame's rows of qmf_buf_real / _imag at the transposer's delay */
  const int32_t *pitch_in_bins; /* [n_ch] or NULL (all 0): frame data pitch_in_bins */
  xaac_hbe_state *state;        /* [n_ch] in/out */
  float *pv_re, *pv_im;         /* [n_ch][32][64]: ph_vocod_qmf_real / _imag rows; bands start_band..end_band-1 written */
  int32_t *status;              /* [n_ch] or NULL: 0, or -1 (parameters outside the tables or rows): such a channel's state
                                   and output are left alone */
} xaac_hbe_apply_batch_desc;

typedef struct xaac_hbe_anal_batch {
  int32_t n_ch;
  xaac_hbe_state *state;        /* [n_ch] in/out: input_buf read, analy_buf, qmf_in_buf rows 12..27 written */
  int32_t *status;              /* [n_ch] or NULL */
} xaac_hbe_anal_batch;

#ifdef __cplusplus
extern "C" {
#endif

/* ixheaacd_real_synth_filt for n_ch channels: one frame's QMF columns -> synth_size real samples per column */
int32_t xaac_hbe_real_synth_batch(xaac_ctx *ctx, const xaac_hbe_synth_batch *batch);
/* ixheaacd_complex_anal_filt for n_ch channels: no_bins / 2 columns of 2 * synth_size complex sub-band samples */
int32_t xaac_hbe_cplx_anal_batch(xaac_ctx *ctx, const xaac_hbe_anal_batch *batch);

/* ixheaacd_qmf_hbe_apply (hbe_trans.c:224-296) for n_ch channels: time-signal shift, synthesis bank, analysis bank,
   stretch-by-2/3/4 products into qmf_out_buf (with the pitch-adaptive cross products when pitch_in_bins / 12 >= 1),
   rotation of the frame's 32 output rows into pv_re / pv_im.  fft_ready mirrors whether the reference's transposer has its FFT
   pointers: while it has not (a new stream; always for synth_size 20, whose case sets none, hbe_trans.c:159-164) the
   reference re-initialises on every call and so clears both delay lines first (:240-248, :124, :174). */
int32_t xaac_hbe_apply_batch(xaac_ctx *ctx, const xaac_hbe_apply_batch_desc *batch);

#ifdef __cplusplus
}
#endif

#endif /* XAAC_HBE_H */
