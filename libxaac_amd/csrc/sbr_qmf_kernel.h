/* sbr_qmf_kernel.h -- launch interface of the SBR QMF kernels (internal). */
#ifndef XAAC_SBR_QMF_KERNEL_H
#define XAAC_SBR_QMF_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_amd.h"
#include "../../include/xaac_sbr.h"

/* waves per workgroup of the generic banks; each wave owns two channel-frames.  One: seven workgroups of 21 KB fit a CU
   where three pairs did, and with a host that keeps two HIP streams busy (bench.py) single-wave workgroups find room
   beside the other stream's kernels -- C3 on two streams 0.477 -> 0.442 ms (one stream: synthesis 119 -> 117 us) */
#ifndef XAAC_QMF_WAVES
#define XAAC_QMF_WAVES 1
#endif
#define XAAC_QMF_BLOCK (64 * XAAC_QMF_WAVES)
/* analysis: 2 x 1312 int16 history (+pad) and a 64 x 65 int32 exchange tile */
#define XAAC_QMF_ANA_LDS_PER_WAVE (2 * 1312 * 2 + 32 + 64 * 65 * 4)
/* synthesis: max(row tile, ring-sample store) ; row tile 64 x (64|128)+1 words, store 2 x 41 x 128 int16 */
#define XAAC_QMF_SYN_LDS_PER_WAVE_LP 21376 /* 2 channels x 41 slots x (128 + 2) ring samples, rounded up */
#define XAAC_QMF_SYN_LDS_PER_WAVE_HQ (64 * 129 * 4)

/* state / qmf / scale are addressed with explicit per-channel strides so that the same kernels serve the
   stand-alone QMF entry points (tight arrays) and the fused SBR path (fields inside xaac_sbr_state, rows
   inside the per-channel 40 x 64 QMF matrix). */
typedef struct XaacQmfAnaParams {
  int32_t n_ch, ch_fac, low_pow, usb, slot_stride;
  int32_t state_stride;   /* bytes between consecutive channels' xaac_qmf_ana_state */
  int32_t qmf_ch_stride;  /* words between consecutive channels' slot 0 */
  const int16_t *pcm;
  xaac_qmf_ana_state *state;
  int32_t *qmf;
  /* fused SBR path, HQ mode: the band limit of the final rotation is per channel -- what
     ixheaacd_rescale_x_overlap (sbrdec_lpfuncs.c:470) leaves in str_codec_qmf_bank.usb before the bank
     runs: frame->max_qmf_subband_aac when the frame is processed, else the state's codec_usb (state then
     points into xaac_sbr_state).  NULL: use `usb` for every channel. */
  const xaac_sbr_frame *frame;
  int32_t *zero_words; /* optional: two words the HQ kernel clears for the launch behind it (the SBR core's counters) */
} XaacQmfAnaParams;

typedef struct XaacQmfSynParams {
  int32_t n_ch, ch_fac, low_pow, lsb, usb, split, slot_stride;
  int32_t state_stride, qmf_ch_stride;
  int32_t scale_stride;   /* int16 words between channels' {lb, ov_lb, hb, st_syn[, lsb, usb]} */
  int32_t per_ch_bands;   /* 1: lsb/usb come from scale[4], scale[5] of each channel */
  int32_t down_sample;    /* 1: the 32-channel bank (sbrdec_initfuncs.c:1165): 1024 samples out, 640-sample ring */
  const int32_t *qmf;
  const int16_t *scale;
  xaac_qmf_syn_state *state;
  int16_t *pcm;
  /* output addressing: 0 -> the ch_fac interleave of the C ABI; else channel c's sample n goes to
     pcm[c * pcm_ch_stride + n * pcm_sample_stride] (one launch per output channel of a PS stream) */
  int32_t pcm_ch_stride, pcm_sample_stride;
  int32_t *dbg; /* profiling builds (-DXS_PROFILE) only: cycle counters at dbg[64..], else unused */
} XaacQmfSynParams;

/* HE-AACv2: both complex synthesis banks of a stream in one wave (channel 0 = left, state in xaac_sbr_state; channel
   1 = right, state in xaac_ps_state), output as interleaved L,R pairs.  scale[c] + 8 i: lb, ov_lb, hb, st_syn scales,
   lsb, usb, and [6] != 0 where the channel is not synthesised this frame (bank and output samples left alone). */
#define XAAC_QMF_SYN_PAIR_LDS (2 * 42 * 65 * 4) /* the 2 x 42 x 65 pair rows; the two 32 x 65-word half-row tiles (one per wave, one channel at a time) alias them */
typedef struct XaacQmfSynPairParams {
  int32_t n;       /* streams */
  int32_t split;   /* first slot of the current frame's low band (op_delay = 6) */
  const int32_t *qmf[2];
  int32_t qmf_stride[2];    /* words between consecutive streams' slot 0 (rows of 64 re | 64 im) */
  const int16_t *scale[2];
  xaac_qmf_syn_state *state[2];
  int32_t state_stride[2];  /* bytes */
  int16_t *pcm;             /* [n][2048][2] */
  int32_t *status;          /* [n] or NULL: -1 where a stream's ring state is refused (see the kernel) */
} XaacQmfSynPairParams;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_qmf_synthesis_pair(const XaacQmfSynPairParams *p, hipStream_t stream);
hipError_t xaac_launch_qmf_analysis(const XaacQmfAnaParams *p, int grid, hipStream_t stream);
#define XAAC_QMF_ELD_LDS (4 * (288 + 512) * 2 + 64 * 65 * 4) /* four channels' time-ordered history + the exchange tile */
/* the LD / ELD banks inside the low-delay SBR chain (xaac_sbr_eld_process_batch): the public batch + where the chain keeps
   things.  All optional: zero = the stand-alone entry points' tight arrays and batch-wide band limits. */
typedef struct XaacQmfEldChain {
  int32_t state_stride;         /* bytes between consecutive channels' bank states (0: the state struct's own size) */
  int32_t pcm_ch_fac;           /* interleave stride of the PCM (0 / 1: planar) */
  const xaac_sbr_frame *frame;  /* analysis: the band limit of the final rotation is per channel -- what ixheaacd_rescale_x_overlap
                                   leaves in str_codec_qmf_bank.usb in front of the bank: the frame's max_qmf_subband_aac when it is
                                   processed, else codec_usb */
  const int16_t *codec_usb;     /* ... of channel 0, the others state_stride bytes apart */
  const int16_t *syn_par;       /* synthesis, [n_ch][8]: lb, ov_lb, hb, st_syn scales, lsb, usb, [6] != 0: channel left alone */
} XaacQmfEldChain;
hipError_t xaac_launch_qmf_analysis_eld(const xaac_qmf_ana_eld_batch *p, hipStream_t stream);
hipError_t xaac_launch_qmf_analysis_eld_chain(const xaac_qmf_ana_eld_batch *p, const XaacQmfEldChain *c, hipStream_t stream);
hipError_t xaac_launch_qmf_synthesis_eld_chain(const xaac_qmf_syn_eld_batch *p, const XaacQmfEldChain *c, hipStream_t stream);
#define XAAC_QMF_ELD_SYN_LDS (4 * 25 * 130 * 2) /* four channels x (9 + 16) slots of ring samples, padded rows */
hipError_t xaac_launch_qmf_synthesis_eld(const xaac_qmf_syn_eld_batch *p, hipStream_t stream);
hipError_t xaac_launch_qmf_synthesis(const XaacQmfSynParams *p, int grid, hipStream_t stream);
int xaac_qmf_blocks_per_cu(int which);
#ifdef __cplusplus
}
#endif
#endif
