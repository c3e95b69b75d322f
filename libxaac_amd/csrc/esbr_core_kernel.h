/* esbr_core_kernel.h -- launch interface of the Path A HF generator / envelope adjuster kernel (internal). */
#ifndef XAAC_ESBR_CORE_KERNEL_H
#define XAAC_ESBR_CORE_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_esbr.h"

#define XAAC_ESBR_OUT_ROWS 42                                   /* 8 history + 32 + 2 rows a VARVAR frame can reach */
#define XAAC_ESBR_L_ROWS 38                                     /* 32 regrouped rows + the 6 look-ahead rows of the PS hybrid filter */
#define XAAC_ESBR_PH_ROWS 40                                    /* ph_vocod_qmf: 8 history rows + the transposer's 32 */
#define XAAC_ESBR_WS_FLOATS (2 * 2048 + 2 * XAAC_ESBR_OUT_ROWS * 64 + 2 * XAAC_ESBR_L_ROWS * 64 + 2 * 2048 + 2 * XAAC_ESBR_PH_ROWS * 64 + 16 * 64) /* analysis rows, sbr_qmf_out, left rows, right rows, transposer rows, the PVC decoder's envelope */

/* 4:1 SBR (64 slots, four to an envelope time slot; op_delay 12): the low-band matrix is a scratch of its own -- 14 history rows,
   the 16-channel bank's 64 rows (written in place by the bank), 2 rows of zeros -- and sbr_qmf_out grows likewise; no PS rows,
   no transposer rows */
#define XAAC_ESBR_Q_ROWS_4_1 80
#define XAAC_ESBR_OUT_ROWS_4_1 82
#define XAAC_ESBR_WS_FLOATS_4_1 (2 * XAAC_ESBR_Q_ROWS_4_1 * 64 + 2 * XAAC_ESBR_OUT_ROWS_4_1 * 64 + 2 * 64 * 64 + 16 * 64)

typedef struct XaacEsbrCoreParams {
  int32_t n_ch;
  const xaac_sbr_header *header;
  const xaac_sbr_frame *frame;
  const xaac_esbr_side *side;
  xaac_esbr_state *state;
  const float *ana_re, *ana_im; /* [n_ch][32][64] this frame's analysis rows */
  float *out_re, *out_im;       /* [n_ch][42][64] scratch: sbr_qmf_out */
  float *syn_re, *syn_im;       /* [n_ch][38][64] regrouped rows for the synthesis bank (+ rows 32..37, bands 0..4, with PS) */

  int32_t with_ps;
  int32_t *status;
  const xaac_hbe_state *hbe;    /* [n_ch] or NULL: the channels' harmonic transposers, already run on this frame */
  float *ph_re, *ph_im;         /* [n_ch][40][64] scratch: ph_vocod_qmf (rows 8..39 written by the transposer) */
  int32_t hbe_lds_synth_size;   /* the transposer launches' LDS hint (hbe_kernel.h): a channel whose bank is larger was not run */
  const xaac_esbr_pvc_side *pvc_side; /* [n_ch] or NULL (with pvc_state, pvc_out): channels with PVC frames (xaac_esbr.h) */
  xaac_esbr_pvc_state *pvc_state;
  float *pvc_out;               /* [n_ch][16][64] scratch: pvc_dec_out_buf */
  int32_t usf4;                 /* 4:1 SBR: out_re / out_im [n_ch][82][64], syn_re / syn_im [n_ch][64][64], and */
  float *q_re, *q_im;           /* [n_ch][80][64]: qmf_buf rows, 14..77 written by the analysis bank (ana_re / ana_im unused) */
  const xaac_hbe_dft_state *dft; /* [n_ch] or NULL, instead of hbe: the channels' DFT transposers (-esbr_hq:1), already run on this
                                    frame (last_status says whether) */
} XaacEsbrCoreParams;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_esbr_core(const XaacEsbrCoreParams *p, hipStream_t stream);

typedef struct XaacEsbrPsParams {
  int32_t n;
  const xaac_sbr_header *header;  /* sub_band_end */
  const xaac_sbr_frame *frame;    /* apply_processing */
  const xaac_ps_frame *ps_frame;
  xaac_esbr_ps_state *ps_state;
  float *l_re, *l_im;             /* [n][38][64] in: regrouped rows; out: the left channel's rows 0..31 */
  float *r_re, *r_im;             /* [n][32][64] out: the right channel */
  int32_t *status;                /* -1 is written where the side info is outside the tool's tables */
} XaacEsbrPsParams;
hipError_t xaac_launch_esbr_ps(const XaacEsbrPsParams *p, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
