/* usac_imdct_kernel.h -- launch interface of the USAC FD IMDCT kernel (internal). */
#ifndef XAAC_USAC_IMDCT_KERNEL_H
#define XAAC_USAC_IMDCT_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_amd.h"

#define XAAC_USAC_WAVES_PER_WG 4
#define XAAC_USAC_LDS (XAAC_USAC_WAVES_PER_WG * 2 * 1024 * 4)

typedef struct XaacUsacImdctParams {
  int32_t n_ch, ccfl; /* ccfl: 1024 or 768 */
  const int32_t *coef;
  const xaac_usac_ics *ics;
  int32_t *overlap;
  uint8_t *shape_prev;
  int32_t *out32;
  float *time;
  int32_t *status;
  const uint8_t *lpd_flags;  /* optional [n_ch]: bit 0 td_frame_prev, bit 1 fac_data_present */
  const xaac_usac_fac *fac;  /* optional [n_ch] */
} XaacUsacImdctParams;

typedef struct XaacUsacFacParams { /* ixheaacd_cal_fac_data for the channels with both LPD flags set */
  int32_t n_ch, ccfl;
  const xaac_usac_ics *ics;
  const uint8_t *lpd_flags;
  const xaac_usac_fac_in *in;
  xaac_usac_fac *out;        /* [n_ch]: data, q; q = XAAC_USAC_FAC_REFUSED where the reference's function returns an error */
} XaacUsacFacParams;
#define XAAC_USAC_FAC_REFUSED 0x7fffffff

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_usac_fac(const XaacUsacFacParams *p, hipStream_t stream);
hipError_t xaac_launch_usac_imdct(const XaacUsacImdctParams *p, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
