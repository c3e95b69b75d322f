/* sbr_ps_kernel.h -- launch interface of the parametric-stereo kernel (internal). */
#ifndef XAAC_SBR_PS_KERNEL_H
#define XAAC_SBR_PS_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_amd.h"
#include "../../include/xaac_sbr.h"

typedef struct XaacPsParams {
  int32_t n;                  /* streams */
  int32_t *x;                 /* [n][40 * 128]: the HQ QMF matrix the core kernel left (slot 0 at row 2);
                                 rows 2..33 become the LEFT channel, in the scale the synthesis bank wants */
  int32_t *xr;                /* [n][32 * 128]: out, the RIGHT channel's 32 slots */
  const xaac_sbr_header *header;   /* [n]: channel_mode */
  const xaac_sbr_frame *sbr_frame; /* [n]: apply_processing */
  const xaac_ps_frame *frame; /* [n] */
  xaac_ps_state *state;       /* [n] */
  xaac_sbr_state *sbr_state;  /* [n]: ps_scale is recorded here */
  int16_t *par_l;             /* [n][8] in: lb, ov_lb, hb, st_syn scales, lsb, usb from the core kernel;
                                 out: the same slots rewritten for the left synthesis launch */
  int16_t *par_r;             /* [n][8] out: scale / band parameters of the right synthesis launch;
                                 [6] = 1 where the stream has no PS this frame (bank and output left alone) */
  int32_t *status;            /* optional [n]: -1 where PS side info had to be clamped into its tables */
  int32_t *dbg;               /* profiling builds (-DXS_PROFILE) only: 32 cycle counters, else unused */
} XaacPsParams;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_sbr_handover(const xaac_sbr_handover_batch *b, hipStream_t stream);
hipError_t xaac_launch_sbr_apply_side(const xaac_sbr_apply_side_batch *b, hipStream_t stream);
hipError_t xaac_launch_ps(const XaacPsParams *p, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
