/*
 * pvc_kernel.hip -- xaac_pvc_process_batch (include/xaac_pvc.h): the PVC envelope decoder, one 64-lane workgroup per channel
 * running pvc.h's team routine (ixheaacd_qmf_enrg_calc, decoder/ixheaacd_sbr_dec.c:80; ixheaacd_pvc_process,
 * decoder/ixheaacd_pred_vec_block.c:176).  A frame moves 1.5 KB of QMF samples in and 4 KB of energies out: the launch is
 * bounded by its latency chain (global loads -> log10 -> three LDS phases -> pow -> stores), not by HBM or the VALU; a
 * batch of 8192 channels fills the chip 32 workgroups deep per CU.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pvc.h"
#include "pvc_kernel.h"

__global__ __launch_bounds__(64) void xaac_pvc_kernel(XaacPvcParams p) {
  const int ch = blockIdx.x;
  __shared__ XpWork w;
  __shared__ xaac_pvc_frame f;
  { /* the frame's 40 bytes as ten words */
    const int32_t *src = reinterpret_cast<const int32_t *>(p.frame + ch);
    if (threadIdx.x < sizeof(xaac_pvc_frame) / 4) reinterpret_cast<int32_t *>(&f)[threadIdx.x] = src[threadIdx.x];
  }
  __syncthreads();
  const XpCx cx = {(int)threadIdx.x, 64};
  const size_t o = (size_t)ch * p.qmf_stride;
  const int rc = xp_process(cx, &w, &f, p.qmf_re + o, p.qmf_im + o, (size_t)p.qmf_stride, p.state + ch, p.out + (size_t)ch * XAAC_PVC_SLOTS * 64);
  if (p.status && threadIdx.x == 0) p.status[ch] = rc;
}

hipError_t xaac_launch_pvc(const XaacPvcParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_pvc_kernel, dim3(p->n_ch), dim3(64), 0, stream, *p);
  return hipGetLastError();
}

/* xaac_warm_up (xaac_abi.cpp): asking for a kernel's attributes puts this translation unit's code object on the device */
extern "C" hipError_t xaac_warm_pvc(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&xaac_pvc_kernel));
}
