/*
 * esbr_ps_kernel.hip -- gfx950 kernel for the float parametric-stereo tool of the reference's default SBR path
 * (ixheaacd_esbr_apply_ps, decoder/ixheaacd_ps_dec_flt.c:389, between the regrouping and the two synthesis banks of
 * ixheaacd_esbr_synthesis_filt_block, sbr_dec.c:447); arithmetic in esbr_ps.h.
 *
 * Mapping: one wave = one stream-frame.  Hybrid analysis / synthesis and the band powers with lane = slot, the transient
 * detector with lane = parameter bin, the decorrelator and the rotation with lane = hybrid sub-band or QMF band (each a
 * recursion over the 32 slots: all-pass rings, delays, a mixing matrix stepped slot by slot).  Hybrid sub-band signals,
 * powers and transient ratios live in LDS (11.6 KB), the two 64-band matrices and the 16 KB of delay-line state in global
 * memory (L2-resident while the wave works on them).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef XE_PROFILE /* tools/prof_esbr_core.py ps: cycles of lane 0 between XE_T hooks, summed over stream-frames */
__shared__ long long xe_prof_last;
__shared__ long long xe_prof_acc[16];
#define XE_T(i)                            \
  do {                                     \
    if (threadIdx.x == 0) {                \
      const long long t_ = clock64();      \
      xe_prof_acc[i] += t_ - xe_prof_last; \
      xe_prof_last = t_;                   \
    }                                      \
  } while (0)
#endif
#include "esbr_ps.h"
#include "esbr_core_kernel.h"

__global__ __launch_bounds__(64, 3) void xaac_esbr_ps_kernel(XaacEsbrPsParams p) {
  __shared__ XfWork w;
  const int n = blockIdx.x, lane = threadIdx.x;
  const XsCx cx = {lane, 64};
#ifdef XE_PROFILE
  if (lane == 0) {
    for (int i = 0; i < 16; i++) xe_prof_acc[i] = 0;
    xe_prof_last = clock64();
  }
  __syncthreads();
#endif
  const xaac_ps_frame *pf = p.ps_frame + n;
  float *lre = p.l_re + (size_t)n * XAAC_ESBR_L_ROWS * 64, *lim = p.l_im + (size_t)n * XAAC_ESBR_L_ROWS * 64;
  float *rre = p.r_re + (size_t)n * 2048, *rim = p.r_im + (size_t)n * 2048;
  const XeMat L = {lre, lim}, R = {rre, rim};
  /* the side info's head (quantiser flags, seven borders, the envelope count: five words), the frame's processing flag and the
     band limit, one load each and all in flight together (read member by member inside the checks they were up to ten memory
     round trips one behind the other) */
  static_assert(offsetof(xaac_ps_frame, border_position) == 4 && offsetof(xaac_ps_frame, num_env) == 18, "layout");
  const int head_v = lane < 5 ? reinterpret_cast<const int32_t *>(pf)[lane] : 0;
  const int apply_v = p.frame[n].apply_processing, sbe_v = p.header[n].sub_band_end, mode_v = p.header[n].channel_mode;
  const auto head16 = [&](int e) { /* element e of the head, a short */
    const int wv = __builtin_amdgcn_readlane(head_v, e >> 1);
    return (int)(int16_t)((e & 1) ? (wv >> 16) : wv);
  };
  int border[XAAC_PS_MAX_ENV + 2];
#pragma unroll
  for (int e = 0; e < XAAC_PS_MAX_ENV + 2; e++) border[e] = head16(2 + e);
  const int num_env = head16(9);
  bool bad = num_env < 1 || num_env > XAAC_PS_MAX_ENV || border[0] < 0;
  if (!bad) {
#pragma unroll
    for (int e = 0; e < XAAC_PS_MAX_ENV; e++) bad |= e < num_env && (border[e] > border[e + 1] || border[e + 1] > 32);
  }
  const int apply = __builtin_amdgcn_readfirstlane(apply_v), sub_band_end = __builtin_amdgcn_readfirstlane(sbe_v);
  /* a stream without parametric stereo in a PS batch (channel_mode != PS_STEREO, as the fixed-point PS kernel reads it): its
     side-info row means nothing and it has no right channel */
  if (__builtin_amdgcn_readfirstlane(mode_v) != 3) return; /* (the right bank's launch skips the stream too; out_r is left as it is) */
  if (apply && !bad) {
    /* The right channel's rows: inside the frame's PS range [border 0, last border) the decorrelator writes every band from 3
       up and the hybrid synthesis bands 0..2 of every row, so only rows outside the range (none, for the borders 0 and 32 an
       encoder sends) have to be cleared -- not 16 KB of zeros per stream that the same kernel then overwrites */
    int k1 = border[1];
#pragma unroll
    for (int e = 2; e <= XAAC_PS_MAX_ENV; e++) k1 = e == num_env ? border[e] : k1;
    const int k0 = border[0];
    for (int i = 0; i < 32; i++)
      if (i < k0 || i >= k1) { /* (uniform) */
        rre[64 * i + lane] = 0.0f;
        rim[64 * i + lane] = 0.0f;
      }
    __syncthreads();
    xf_apply_ps(cx, pf, p.ps_state + n, &w, L, R, sub_band_end);
#ifdef XE_PROFILE
    if (lane < 16) atomicAdd(reinterpret_cast<unsigned long long *>(p.status) + 16 + lane, (unsigned long long)xe_prof_acc[lane]);
#endif
  } else { /* no SBR processing this frame: the right channel is the left one (sbr_dec.c:516-523) */
    for (int i = 0; i < 32; i++) {
      rre[64 * i + lane] = lre[64 * i + lane];
      rim[64 * i + lane] = lim[64 * i + lane];
    }
    if (lane == 0 && bad && apply && p.status) p.status[n] = -1;
  }
}

extern "C" hipError_t xaac_launch_esbr_ps(const XaacEsbrPsParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_esbr_ps_kernel, dim3(p->n), dim3(64), 0, stream, *p);
  return hipGetLastError();
}

/* xaac_warm_up (xaac_abi.cpp): asking for a kernel's attributes puts this translation unit's code object on the device */
extern "C" hipError_t xaac_warm_esbr_ps(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&xaac_esbr_ps_kernel));
}
