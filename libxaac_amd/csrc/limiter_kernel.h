/* limiter_kernel.h -- launch interface of the peak limiter kernel (internal). */
#ifndef XAAC_LIMITER_KERNEL_H
#define XAAC_LIMITER_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_amd.h"

typedef struct XaacLimiterParams {
  int32_t n_streams, frame_len, num_channels;
  int32_t planar; /* samples: [channel][frame_len] per stream instead of [frame_len][channel] */
  int32_t *samples;
  int64_t stride;
  const int8_t *qshift_adj;
  xaac_limiter_state *state;
  int16_t *pcm16;
  int32_t *status;
  float *ws_gain;   /* [n_streams][1024]: target gains -> gains of the streams the recursion runs on */
  int32_t *ws_flag; /* [n_streams][2]: finished by the front kernel?, first sample of the recursion */
  long long *dbg; /* phase timers (profiling builds) */
} XaacLimiterParams;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_limiter(const XaacLimiterParams *p, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
