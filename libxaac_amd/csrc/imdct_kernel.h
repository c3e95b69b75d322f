/* imdct_kernel.h -- launch interface between the C-ABI layer (xaac_abi.cpp) and
 * the gfx950 kernel (imdct_kernel.hip).  Internal; the public ABI is
 * include/xaac_amd.h. */
#ifndef XAAC_IMDCT_KERNEL_H
#define XAAC_IMDCT_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_amd.h"

#ifndef XAAC_IMDCT_WAVES
#define XAAC_IMDCT_WAVES 4                       /* independent waves per workgroup */
#endif
#define XAAC_IMDCT_BLOCK (64 * XAAC_IMDCT_WAVES)
#ifndef XAAC_IMDCT_MIN_WAVES_PER_SIMD
#define XAAC_IMDCT_MIN_WAVES_PER_SIMD 4         /* register budget: <= 128 VGPRs */
#endif
#define XAAC_IMDCT_LDS_WIN_BYTES 4608            /* 2x1024 + 2x128 int16 windows */
#define XAAC_IMDCT_LDS_WAVE_WORDS (1024 + 512)   /* exchange tile + old-overlap copy */
#define XAAC_IMDCT_LDS_CONST_WORDS (1024 + 4 * 448) /* rotation pairs + split pass-2/3 twiddles, lane-major */
#define XAAC_IMDCT_LDS_BYTES \
  (XAAC_IMDCT_LDS_WIN_BYTES + 4 * (XAAC_IMDCT_LDS_CONST_WORDS + XAAC_IMDCT_WAVES * XAAC_IMDCT_LDS_WAVE_WORDS))

enum { XAAC_K_ONLY_LONG = 0, XAAC_K_LONG_START = 1, XAAC_K_EIGHT_SHORT = 2, XAAC_K_LONG_STOP = 3 };

typedef struct XaacImdctParams {
  int32_t n_ch;
  int32_t ch_fac;
  const int32_t *spec;
  const xaac_ics_info *ics;
  int32_t *overlap;
  xaac_ovl_state *state;
  int32_t *out32;
  int16_t *pcm16;
  int8_t *qshift_adj;
  int32_t pcm_mode;
  int32_t *status;
} XaacImdctParams;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_imdct(const XaacImdctParams *p, int grid, hipStream_t stream);
int xaac_imdct_blocks_per_cu(void);
#ifdef __cplusplus
}
#endif
#endif
