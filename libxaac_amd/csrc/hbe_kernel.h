/* hbe_kernel.h -- launch interface of the harmonic transposer's polyphase bank kernels (internal). */
#ifndef XAAC_HBE_KERNEL_H
#define XAAC_HBE_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_hbe.h"

#define XAAC_HBE_SYN_LDS ((41 * 40 + 32 * 265) * 4) /* v of 9 + 32 columns; 32 lanes' transform scratch (odd stride) */
#define XAAC_HBE_ANA_LDS ((16 * 80 + 16 * 80 + 16 * 513) * 4) /* u and results of 16 columns; 16 lanes' scratch */

typedef struct XaacHbeSynParams {
  int32_t n_ch, num_columns;
  const float *qmf_re, *qmf_im; /* [n_ch][num_columns][64] */
  xaac_hbe_state *state;        /* [n_ch] */
  int32_t *status;              /* [n_ch] or NULL */
  const int32_t *pitch;         /* apply mode: [n_ch] or NULL */
  int32_t apply;                /* 1: as the first step of ixheaacd_qmf_hbe_apply (time-signal shift, the re-initialisation
                                   while fft_ready is 0, the frame's parameter check) */
} XaacHbeSynParams;

typedef struct XaacHbeAnaParams {
  int32_t n_ch;
  xaac_hbe_state *state;
  int32_t *status;
  const int32_t *pitch;
  int32_t apply;                /* 1: second step of the apply chain (qmf_in_buf rows moved down first) */
} XaacHbeAnaParams;

#define XAAC_HBE_POST_THREADS 256
#define XAAC_HBE_POST_LDS (256 * 25 * 4) /* the blocks (+ cross terms) of 16 bands x 16 columns */
typedef struct XaacHbePostParams {
  int32_t n_ch;
  xaac_hbe_state *state;
  const int32_t *pitch;
  float *pv_re, *pv_im;         /* [n_ch][32][64] */
} XaacHbePostParams;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_hbe_synth(const XaacHbeSynParams *p, hipStream_t stream);
hipError_t xaac_launch_hbe_anal(const XaacHbeAnaParams *p, hipStream_t stream);
hipError_t xaac_launch_hbe_post(const XaacHbePostParams *p, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
