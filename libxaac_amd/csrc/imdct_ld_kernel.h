/* imdct_ld_kernel.h -- launch interface of the AAC-LD / ELD IMDCT kernels (internal). */
#ifndef XAAC_IMDCT_LD_KERNEL_H
#define XAAC_IMDCT_LD_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_amd.h"

#define XAAC_LD_WAVES_PER_WG 4
/* per wave: the transform's two arrays (1024 + 512 words) and the old overlap (ELD: 3 x frame_length, LD: frame_length / 2) */
#define XAAC_LD_LDS(frame_length, eld) (XAAC_LD_WAVES_PER_WG * (1024 + 512 + ((eld) ? 3 * (frame_length) : (frame_length) / 2)) * 4)

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_imdct_ld(const xaac_imdct_ld_batch *p, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
