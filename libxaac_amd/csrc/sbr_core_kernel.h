/* sbr_core_kernel.h -- launch interface of the SBR core kernel (internal). */
#ifndef XAAC_SBR_CORE_KERNEL_H
#define XAAC_SBR_CORE_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_amd.h"

#define XAAC_SBR_X_ROWS 40                       /* 2 LPC history rows + 6 overlap slots + 32 new slots */
#define XAAC_SBR_X_WORDS (XAAC_SBR_X_ROWS * 64)  /* int32 words of one channel's QMF matrix */
#define XAAC_SBR_NARROW_BANDS 48                 /* bands per LDS row of the HQ core's narrow-row kernel */
#ifndef XAAC_SBR_CORE_HQ_WAVES
#define XAAC_SBR_CORE_HQ_WAVES 4                 /* its waves per workgroup (they share the lookup tables in LDS) */
#endif

typedef struct XaacSbrCoreParams {
  int32_t n_ch;
  const xaac_sbr_header *header;
  const xaac_sbr_frame *frame;
  xaac_sbr_state *state;
  int32_t *x;        /* [n_ch][XAAC_SBR_X_WORDS] */
  int16_t *syn_par;  /* [n_ch][8]: lb, ov_lb, hb, st_syn scales, synthesis lsb, usb */
  int32_t *status;   /* optional [n_ch] */
  /* HQ only, optional (both or neither): [n_ch] stream numbers + one counter.  With them the launch runs the narrow-row
     kernel first and the streams it cannot take through the 64-band rows afterwards (sbr_core_kernel.hip) */
  int32_t *defer_list, *defer_count;
  int32_t *work_counter; /* = defer_count + 1: the persistent waves' next channel-frame */
  int32_t num_cu;        /* compute units of the device (grid of the persistent launch) */
  int32_t counters_zeroed; /* 1: an earlier launch on the stream has cleared defer_count / work_counter */
  int32_t narrow_only;     /* the caller's assertion (xaac_sbr_hq_batch.max_band_hint): no list launch; a stream that needs the 64-band
                              rows is refused */
} XaacSbrCoreParams;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_sbr_core_lp(const XaacSbrCoreParams *p, hipStream_t stream);
/* HQ: x rows are 128 words (64 real | 64 imaginary): [n_ch][2 * XAAC_SBR_X_WORDS] */
hipError_t xaac_launch_sbr_core_hq(const XaacSbrCoreParams *p, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
