/* esbr_core_kernel.h -- launch interface of the Path A HF generator / envelope adjuster kernel (internal). */
#ifndef XAAC_ESBR_CORE_KERNEL_H
#define XAAC_ESBR_CORE_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_esbr.h"

#define XAAC_ESBR_OUT_ROWS 42                                   /* 8 history + 32 + 2 rows a VARVAR frame can reach */
#define XAAC_ESBR_WS_FLOATS (2 * 2048 + 2 * XAAC_ESBR_OUT_ROWS * 64 + 2 * 2048) /* analysis rows, sbr_qmf_out, regrouped rows */

typedef struct XaacEsbrCoreParams {
  int32_t n_ch;
  const xaac_sbr_header *header;
  const xaac_sbr_frame *frame;
  const xaac_esbr_side *side;
  xaac_esbr_state *state;
  const float *ana_re, *ana_im; /* [n_ch][32][64] this frame's analysis rows */
  float *out_re, *out_im;       /* [n_ch][42][64] scratch: sbr_qmf_out */
  float *syn_re, *syn_im;       /* [n_ch][32][64] regrouped rows for the synthesis bank */
  int32_t *status;
} XaacEsbrCoreParams;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_esbr_core(const XaacEsbrCoreParams *p, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
