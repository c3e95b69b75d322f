/*
 * xaac_hbe.h -- boundary formats of the QMF-domain harmonic transposer ("HBE") of the eSBR tool: the per-channel state
 * of ia_esbr_hbe_txposer_struct (decoder/ixheaacd_sbr_dec.h:30-100) that ixheaacd_qmf_hbe_apply
 * (decoder/ixheaacd_hbe_trans.c:224) keeps between frames, and batched entry points for its two polyphase banks
 * (decoder/ixheaacd_esbr_polyphase.c):
 *
 *   xaac_hbe_real_synth_batch  <-> ixheaacd_real_synth_filt     (esbr_polyphase.c:157-274)
 *   xaac_hbe_cplx_anal_batch   <-> ixheaacd_complex_anal_filt   (esbr_polyphase.c:48-155)
 *   xaac_hbe_apply_batch       <-> ixheaacd_qmf_hbe_apply       (hbe_trans.c:224-296)
 *   xaac_hbe_dft_anal_batch_run <-> ixheaacd_dft_hbe_cplx_anal_filt (esbr_polyphase.c:276-338; the DFT transposer's bank)
 *   xaac_hbe_dft_apply_batch_run <-> ixheaacd_dft_hbe_apply      (hbe_dft_trans.c:771-941; -esbr_hq:1)
 *
 * Scope: the QMF transposer (esbr_hq = 0) at 2:1 SBR of 1024-sample cores: no_bins = 32 QMF columns per frame, bank
 * sizes synth_size = 4, 8, 12, 16, 20 (hbe_trans.c:111-112), and the DFT transposer (esbr_hq = 1) at the sizes its
 * transforms exist for.  All samples are FLOAT32; the QMF transposer's and the banks' results are bit-identical to the
 * reference's x86-64 build (float operations in the reference's order, no contraction); the DFT transposer's agree to float
 * rounding (see below).
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
  int32_t max_synth_size;       /* optional hint as xaac_esbr_sbr_batch::hbe_max_synth_size: 4, 8, or 0 = any */
} xaac_hbe_apply_batch_desc;

/* The DFT transposer's (esbr_hq) analysis bank, ixheaacd_dft_hbe_cplx_anal_filt: per channel its delay line and the two
   sizes ixheaacd_dft_hbe_data_reinit derives (hbe_dft_trans.c:318-321) */
typedef struct xaac_hbe_dft_anal_state {
  float analy_buf[640];         /* analy_buf: 10 * analy_size samples in use */
  int32_t analy_size;           /* 4 .. 64, a multiple of 4 */
  int32_t a_start;              /* first sub-band written; a_start + analy_size <= 64 */
} xaac_hbe_dft_anal_state;

typedef struct xaac_hbe_dft_anal_batch {
  int32_t n_ch;
  int32_t no_bins;              /* columns per frame, <= 32 */
  const float *time_in;         /* [n_ch][in_stride]: ptr_output_buf (column idx reads samples idx * analy_size + 1 ..) */
  int32_t in_stride;            /* floats per channel, >= no_bins * analy_size + 1 */
  const float *coef_re, *coef_im; /* [n_cfg][64][128]: str_dft_hbe_anal_coeff.real / .imag of the configurations in use */
  const int32_t *cfg;           /* [n_ch]: which configuration a channel uses, or NULL (all 0) */
  xaac_hbe_dft_anal_state *state; /* [n_ch] in/out */
  float *qmf_re, *qmf_im;       /* [n_ch][no_bins + 2][64] in/out: qmf_buf_real / _imag rows.  Real rows: sub-bands a_start..63
                                   rewritten.  Imaginary rows: the reference clears 128 floats from [idx][a_start] on, i.e. into
                                   the next two rows (esbr_polyphase.c:300-301): rows 1..no_bins come out zero outside the
                                   written sub-bands, row 0 keeps its sub-bands below a_start, row no_bins + 1 loses them */
  int32_t *status;              /* [n_ch] or NULL */
} xaac_hbe_dft_anal_batch;

/* ---- the DFT transposer itself (-esbr_hq:1): ixheaacd_dft_hbe_apply, decoder/ixheaacd_hbe_dft_trans.c:771-941 -------------
 * Per frame: the time-signal shift and the real synthesis bank (esbr_polyphase.c:157, its esbr_hq layout), then eight hops of
 * window -> real FFT -> polar form -> stretch by 2, 3, 4 with the pitch-adaptive cross products (:576-769) -> inverse real FFT
 * -> window -> overlap-add, then the analysis bank above.  FLOAT32 samples; atan2 / cos / sin / pow / cbrt / sqrt are the
 * device's double-precision ones where the reference calls glibc's, and the transforms are this library's own (two-pass
 * Cooley-Tukey, 16 x N/16) where the reference has hand-unrolled ones: results agree with the reference to float rounding
 * (tests: relative 2e-5 of the frame's peak), not bit for bit -- the tolerance BASELINE.json's north_star gives float SBR
 * is +-1 LSB of the 16-bit PCM, which the drop-in test holds the decoded streams to.
 * Bank sizes: the transforms the reference has (hbe_dft_trans.c:508-549): synth_size 12 or 16 (8 with oversampling),
 * analy_size 28, 32 or (without oversampling) 48; anything else is refused (status -1, state left alone but for last_status) where the reference fails the
 * frame. */
#define XAAC_HBE_DFT_MAX_ANA 512 /* ana_fft_size[0] = 32 * synth_size */
#define XAAC_HBE_DFT_MAX_SYN 768 /* syn_fft_size[0] = 16 * analy_size */
typedef struct xaac_hbe_dft_state {
  float input_buf[2 * XAAC_HBE_DFT_MAX_ANA];  /* ptr_input_buf: 2 * ana_fft_size[0] samples in use */
  float output_buf[4 * XAAC_HBE_DFT_MAX_SYN]; /* ptr_output_buf: 4 * syn_fft_size[0] samples in use */
  float synth_buf[1280];                      /* the real synthesis bank's delay line */
  xaac_hbe_dft_anal_state anal;               /* the analysis bank's delay line, analy_size, a_start */
  int32_t synth_size, k_start;                /* hbe_dft_trans.c:287-288 */
  int32_t start_band, end_band;               /* :283-284 (not read by the transposer itself) */
  int32_t max_stretch;                        /* 2 .. 4 */
  int32_t x_over_qmf[6];                      /* the patches' cross-over bands (:399-445): not read by the transposer; the envelope
                                                 adjuster's limiter bands take them inside the Path A chain (xaac_esbr.h) */
  int32_t last_status;                        /* written by every call: 0, or -1 where the sizes had no transform (the chain's
                                                 later stages read it) */
} xaac_hbe_dft_state;

/* What ixheaacd_dft_hbe_data_reinit derives from the SBR frequency tables besides the sizes above: the two time windows and
   the frequency-domain cross-over windows of the (up to three) patches, for fft_size 1024 and 1536 (oversampling).  Made
   on the host when a header resets the SBR decoder; channels of one configuration share one. */
typedef struct xaac_hbe_dft_cfg {
  float anal_window[XAAC_HBE_DFT_MAX_ANA];  /* anal_window: ana_fft_size[0] coefficients */
  float synth_window[XAAC_HBE_DFT_MAX_SYN]; /* synth_window: syn_fft_size[0] coefficients */
  float fd_win[3][2][772];                  /* fd_win_buf[trans_fac - 2][oversampling][0 .. fft_size / 2] */
} xaac_hbe_dft_cfg;

typedef struct xaac_hbe_dft_apply_batch {
  int32_t n_ch;
  const float *qmf_re, *qmf_im;    /* [n_ch][32][64]: the frame's rows of qmf_buf_real / _imag at the transposer's delay */
  const int32_t *pitch_in_bins;    /* [n_ch] or NULL (all 0) */
  const int32_t *oversampling;     /* [n_ch] or NULL (all 0): the frame's over_sampling_flag (sbr_dec.c:884) */
  const xaac_hbe_dft_cfg *cfg_tab; /* [n_cfg] */
  const float *coef_re, *coef_im;  /* [n_cfg][64][128]: str_dft_hbe_anal_coeff, as in xaac_hbe_dft_anal_batch */
  const int32_t *cfg;              /* [n_ch] or NULL (all 0) */
  xaac_hbe_dft_state *state;       /* [n_ch] in/out */
  float *pv_re, *pv_im;            /* [n_ch][34][64] in/out: ph_vocod_qmf_real / _imag rows as xaac_hbe_dft_anal_batch.qmf_re / _im */
  int32_t *status;                 /* [n_ch]: 0, or -1 (sizes outside the reference's transforms): state (but last_status) and rows untouched */
  int32_t rows32;                  /* 1: pv_re / pv_im are [n_ch][32][64] blocks that need not hold anything on entry: the clears the
                                      reference's bank makes beyond its 32 rows are left out and sub-bands below a_start are written as
                                      zeros (what the reference's buffer holds there: nothing else ever writes them) -- for a host
                                      that keeps ph_vocod_qmf's rows in an array of its own */
} xaac_hbe_dft_apply_batch;

typedef struct xaac_hbe_anal_batch {
  int32_t n_ch;
  xaac_hbe_state *state;        /* [n_ch] in/out: input_buf read, analy_buf, qmf_in_buf rows 12..27 written */
  int32_t *status;              /* [n_ch] or NULL */
} xaac_hbe_anal_batch;

#ifdef __cplusplus
extern "C" {
#endif

/* ixheaacd_real_synth_filt for n_ch channels: one frame's QMF columns -> synth_size real samples per column */
XAAC_API int32_t xaac_hbe_real_synth_batch(xaac_ctx *ctx, const xaac_hbe_synth_batch *batch);
/* ixheaacd_complex_anal_filt for n_ch channels: no_bins / 2 columns of 2 * synth_size complex sub-band samples */
XAAC_API int32_t xaac_hbe_cplx_anal_batch(xaac_ctx *ctx, const xaac_hbe_anal_batch *batch);

/* ixheaacd_dft_hbe_cplx_anal_filt (esbr_polyphase.c:276-338) for n_ch channels */
XAAC_API int32_t xaac_hbe_dft_anal_batch_run(xaac_ctx *ctx, const xaac_hbe_dft_anal_batch *batch);

/* ixheaacd_dft_hbe_apply (hbe_dft_trans.c:771-941) for n_ch channels */
XAAC_API int32_t xaac_hbe_dft_apply_batch_run(xaac_ctx *ctx, const xaac_hbe_dft_apply_batch *batch);

/* ixheaacd_qmf_hbe_apply (hbe_trans.c:224-296) for n_ch channels: time-signal shift, synthesis bank, analysis bank,
   stretch-by-2/3/4 products into qmf_out_buf (with the pitch-adaptive cross products when pitch_in_bins / 12 >= 1),
   rotation of the frame's 32 output rows into pv_re / pv_im.  fft_ready mirrors whether the reference's transposer has its FFT
   pointers: while it has not (a new stream; always for synth_size 20, whose case sets none, hbe_trans.c:159-164) the
   reference re-initialises on every call and so clears both delay lines first (:240-248, :124, :174). */
XAAC_API int32_t xaac_hbe_apply_batch(xaac_ctx *ctx, const xaac_hbe_apply_batch_desc *batch);

#ifdef __cplusplus
}
#endif

#endif /* XAAC_HBE_H */
