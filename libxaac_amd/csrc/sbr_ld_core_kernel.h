/* sbr_ld_core_kernel.h -- launch interface of the low-delay SBR core kernel (internal). */
#ifndef XAAC_SBR_LD_CORE_KERNEL_H
#define XAAC_SBR_LD_CORE_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_amd.h"

typedef struct XaacSbrLdCoreParams {
  int32_t n_ch, n_slots;
  const xaac_sbr_header *header;
  const xaac_sbr_frame *frame;
  xaac_sbr_eld_state *state;
  int32_t *x;        /* [n_ch][n_slots][128]: the analysed slots in, the synthesis bank's rows out */
  int16_t *syn_par;  /* [n_ch][8]: lb, ov_lb, hb, st_syn scales, synthesis lsb, usb, 1 = not synthesised, 0 */
  int32_t *status;   /* optional [n_ch] */
} XaacSbrLdCoreParams;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_sbr_ld_core(const XaacSbrLdCoreParams *p, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
