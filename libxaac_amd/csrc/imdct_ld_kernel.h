/* imdct_ld_kernel.h -- launch interface of the AAC-LD / ELD IMDCT kernels (internal). */
#ifndef XAAC_IMDCT_LD_KERNEL_H
#define XAAC_IMDCT_LD_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_amd.h"

#define XAAC_LD_WAVES_PER_WG 4
/* per wave: the transform's two arrays (1024 + 512 words); LD's frame_length / 2 old overlap words move into the second one
   once the transform is done, ELD's 3 x frame_length are read where they lie */
#define XAAC_LD_LDS(frame_length, eld) (XAAC_LD_WAVES_PER_WG * (1024 + 512) * 4)

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_imdct_ld(const xaac_imdct_ld_batch *p, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
