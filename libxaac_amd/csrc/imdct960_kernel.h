/* imdct960_kernel.h -- launch interface of the 960-line AAC IMDCT kernel (internal). */
#ifndef XAAC_IMDCT960_KERNEL_H
#define XAAC_IMDCT960_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_amd.h"

#define XAAC_I960_WAVES_PER_WG 4
#define XAAC_I960_LDS (XAAC_I960_WAVES_PER_WG * (960 + 960) * 4)

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_imdct960(const xaac_imdct_batch *p, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
