/* hbe_kernel.h -- launch interface of the harmonic transposer's polyphase bank kernels (internal). */
#ifndef XAAC_HBE_KERNEL_H
#define XAAC_HBE_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_esbr.h"
#include "../../include/xaac_hbe.h"

#define XAAC_HBE_SYN_LDS (41 * 40 * 4) /* v of 9 + 32 columns */
#define XAAC_HBE_ANA_LDS ((16 * 80 + 16 * 80) * 4) /* u and results of 16 columns (the transforms' scratch is private) */

typedef struct XaacHbeSynParams {
  int32_t n_ch, num_columns;
  const float *qmf_re, *qmf_im; /* [n_ch][num_columns][64] */
  xaac_hbe_state *state;        /* [n_ch] */
  int32_t *status;              /* [n_ch] or NULL */
  const int32_t *pitch;         /* apply mode: [n_ch] or NULL */
  int32_t apply;                /* 1: as the first step of ixheaacd_qmf_hbe_apply (time-signal shift, the re-initialisation
                                   while fft_ready is 0, the frame's parameter check) */
  /* inside the Path A chain (xaac_esbr_sbr_process_batch): the pitch comes from the side info and a channel whose
     frame has no SBR processing is skipped (sbr_dec.c:882); NULL elsewhere */
  const xaac_sbr_frame *frame;
  const xaac_esbr_side *side;
  int32_t in_stride;            /* floats between consecutive channels' qmf rows (2048 unless the chain hands in its own) */
} XaacHbeSynParams;

typedef struct XaacHbeAnaParams {
  int32_t n_ch;
  xaac_hbe_state *state;
  int32_t *status;
  const int32_t *pitch;
  int32_t apply;                /* 1: second step of the apply chain (qmf_in_buf rows moved down first) */
  const xaac_sbr_frame *frame;  /* as in XaacHbeSynParams */
  const xaac_esbr_side *side;
} XaacHbeAnaParams;

#define XAAC_HBE_POST_THREADS 256
#define XAAC_HBE_POST_LDS (256 * 25 * 4 + 5 * 16 * 32 * 8) /* the blocks (+ cross terms) of 16 bands x 16 columns; five planes of normalised samples */
typedef struct XaacHbePostParams {
  int32_t n_ch;
  xaac_hbe_state *state;
  const int32_t *pitch;
  float *pv_re, *pv_im;         /* [n_ch][32][64] */
  const xaac_sbr_frame *frame;  /* as in XaacHbeSynParams */
  const xaac_esbr_side *side;
  int32_t pv_stride;            /* floats between consecutive channels' output rows (2048 standalone) */
  int32_t zero_outside;         /* 1: bands outside start_band .. end_band - 1 of the 32 rows are written as zeros (the chain's
                                   scratch rows; the reference leaves whatever its buffer held) */
} XaacHbePostParams;

#define XAAC_HBE_DFT_LDS (32 * 128 * 4) /* u of 32 columns */
typedef struct XaacHbeDftParams {
  int32_t n_ch, no_bins;
  const float *time_in;
  int32_t in_stride;
  const float *coef_re, *coef_im;
  const int32_t *cfg;
  xaac_hbe_dft_anal_state *state;
  float *qmf_re, *qmf_im;
  int32_t *status;
} XaacHbeDftParams;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_hbe_dft_anal(const XaacHbeDftParams *p, hipStream_t stream);
hipError_t xaac_launch_hbe_synth(const XaacHbeSynParams *p, hipStream_t stream);
hipError_t xaac_launch_hbe_anal(const XaacHbeAnaParams *p, hipStream_t stream);
hipError_t xaac_launch_hbe_post(const XaacHbePostParams *p, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
