/* esbr_qmf_kernel.h -- launch interface of the eSBR ("Path A") QMF bank kernels (internal). */
#ifndef XAAC_ESBR_QMF_KERNEL_H
#define XAAC_ESBR_QMF_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_amd.h"

#define XAAC_ESBR_ANA_LDS (64 * 65 * 4)                        /* two channels' time-ordered history (2 x 1312 words), then the 64 x 65 exchange tile in its place */
#define XAAC_ESBR_SYN_LDS (41 * 129 * 4)                       /* ring samples of one channel x (9 + 32) slots; the half-row tile fits */

typedef struct XaacEsbrAnaParams {
  int32_t n_ch;
  const float *core;            /* [n_ch][1024] */
  xaac_esbr_ana_state *state;   /* [n_ch] */
  float *qmf_re, *qmf_im;       /* [n_ch][32][64]; bands 0..31 written */
  int32_t state_stride;         /* bytes between consecutive channels' states */
} XaacEsbrAnaParams;

typedef struct XaacEsbrAnaNbParams { /* the 24- / 16-channel banks of 8:3 / 4:1 SBR */
  int32_t n_ch, nb, n_slots;    /* nb 24 | 16; n_slots <= 64, nb * n_slots <= 1024 */
  const float *core;            /* [n_ch][core_stride], nb * n_slots samples used */
  int32_t core_stride;
  xaac_esbr_ana_state *state;   /* [n_ch]; ring[0 .. 10 nb - 1] used */
  int32_t state_stride;         /* bytes between consecutive channels' states */
  float *qmf_re, *qmf_im;       /* [n_ch][out_stride]: n_slots rows of 64, bands 0..nb-1 written, nb..31 zeroed */
  int32_t out_stride;
} XaacEsbrAnaNbParams;

typedef struct XaacEsbrSynParams {
  int32_t n_ch;
  const float *qmf_re, *qmf_im; /* [n_ch][32][64] */
  xaac_esbr_syn_state *state;   /* [n_ch] */
  float *out;                   /* [n_ch][2048] */
  int32_t state_stride;         /* bytes between consecutive channels' states */
  int32_t in_stride;            /* floats between consecutive channels' row blocks (>= 2048) */
  const xaac_sbr_header *only_ps; /* optional [n_ch]: channels whose channel_mode is not PS_STEREO are left alone (right bank) */
  int32_t out_stride;           /* floats between consecutive channels' output rows; 0 = 2048 */
} XaacEsbrSynParams;

typedef struct XaacEsbrCoreInParams {
  int32_t n_ch, ch_fac;         /* n_ch: channel-frames in total (a multiple of ch_fac) */
  const int16_t *pcm;           /* [n_ch / ch_fac][1024][ch_fac] */
  float *core;                  /* [n_ch][1024] */
} XaacEsbrCoreInParams;

typedef struct XaacEsbrPcmOutParams {
  int32_t n, stride;            /* streams; floats between consecutive streams' planes */
  const float *left, *right;
  int16_t *pcm;                 /* [n][2048][2] */
} XaacEsbrPcmOutParams;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_esbr_core_from_pcm16(const XaacEsbrCoreInParams *p, hipStream_t stream);
hipError_t xaac_launch_esbr_pcm16_from_float(const XaacEsbrPcmOutParams *p, hipStream_t stream);
hipError_t xaac_launch_esbr_analysis(const XaacEsbrAnaParams *p, hipStream_t stream);
hipError_t xaac_launch_esbr_analysis_nb(const XaacEsbrAnaNbParams *p, hipStream_t stream);
hipError_t xaac_launch_esbr_synthesis(const XaacEsbrSynParams *p, hipStream_t stream);
hipError_t xaac_launch_esbr_synthesis_ds(const XaacEsbrSynParams *p, hipStream_t stream); /* 32 synthesis channels: out [n_ch][out_stride | 1024] */
#ifdef __cplusplus
}
#endif
#endif
