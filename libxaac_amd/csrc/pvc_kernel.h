/* pvc_kernel.h -- launch interface of pvc_kernel.hip (internal to the library) */
#ifndef XAAC_PVC_KERNEL_H
#define XAAC_PVC_KERNEL_H

#include <hip/hip_runtime.h>

#include "../../include/xaac_pvc.h"

typedef struct XaacPvcParams {
  int32_t n_ch;
  const xaac_pvc_frame *frame;
  const float *qmf_re, *qmf_im;
  int32_t qmf_stride;
  xaac_pvc_state *state;
  float *out;
  int32_t *status;
} XaacPvcParams;

hipError_t xaac_launch_pvc(const XaacPvcParams *p, hipStream_t stream);

#endif
