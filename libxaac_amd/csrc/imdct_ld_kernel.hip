/*
 * imdct_ld_kernel.hip -- gfx950 kernels for the 512 / 480-line AAC-LD and AAC-ELD IMDCT + windowing + overlap-add (the
 * frame_length 512 / 480 branches of ixheaacd_imdct_process, decoder/ixheaacd_lpfuncs.c:385-486, :804-1010); stages and
 * arithmetic in imdct_ld.h.
 *
 * Mapping as for the 960-line kernel: one wave = one channel-frame, four per workgroup; lines and old overlap read once
 * into LDS with the block exponent as a wave OR; the 256-point transform is four passes of 64 butterflies (one per lane),
 * the 15 x 16 one five short stages; ELD's four-fold output is never materialised (the window stage reads the 2 F
 * transform outputs with the sign / copy map of lpfuncs.c:401-408); PCM16 and the new overlap go straight to global memory.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "imdct_ld.h"
#include "imdct_ld_kernel.h"

namespace {
__device__ __forceinline__ int32_t wave_or(int32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
  return v;
}
/* xl_eld_overlap_add (imdct_ld.h) with the 3 F old overlap words in registers: ov1[i] = word lane + 64 i (what output sample
   lane + 64 i adds), ov2[i] = word F + lane + 64 i (what new overlap word lane + 64 i adds), fetched at the top of the kernel
   with the lines.  Read where they lie -- in place, between the stores of the new words -- every one of the 30 loop steps
   waited for its own load (the bank spent 0.7 of its wave cycles waiting). */
template <int F>
__device__ __forceinline__ void eld_overlap_add_regs(const int32_t *out, const int32_t *ov1, const int32_t *ov2, int32_t *ov_new,
                                                     int16_t *pcm, int stride, int q_shift, int lane) {
  constexpr int delay = F / 4;
  const int16_t *win = F == 512 ? xaac_ld_win_eld_512 : xaac_ld_win_eld_480;
  const int q = q_shift + 2;
#pragma unroll
  for (int i = 0; i < (F + 63) / 64; i++) {
    const int n = lane + 64 * i;
    if (n < F) {
      const int32_t w = fx_mul32x16(xl_eld_z<F>(out, delay + n), win[delay + n]);
      const int32_t v = fx_add_sat(q >= 0 ? fx_shl(w, q) : fx_shr(w, -q), ov1[i]);
      pcm[stride * n] = fx_round16(q >= 0 ? fx_shl_sat(v, 1) : fx_shl(v, 1));
    }
  }
#pragma unroll
  for (int i = 0; i < (3 * F - delay + 63) / 64; i++) {
    const int k = lane + 64 * i;
    if (k < 3 * F - delay) {
      const int32_t w = fx_mul32x16(xl_eld_z<F>(out, delay + F + k), win[delay + F + k]);
      const int32_t sh = q >= 0 ? fx_shl(w, q) : fx_shr(w, -q);
      ov_new[k] = (i < (2 * F + 63) / 64 && k < 2 * F) ? fx_add_sat(sh, ov2[i < (2 * F + 63) / 64 ? i : 0]) : sh;
    }
  }
}
}  // namespace

template <int F, bool ELD>
__global__ __launch_bounds__(64 * XAAC_LD_WAVES_PER_WG) void xaac_imdct_ld_kernel(xaac_imdct_ld_batch p) {
  extern __shared__ __attribute__((aligned(16))) int32_t smem[];
  constexpr int NOV = ELD ? 3 * F : F / 2, PER_WAVE = 1024 + 512;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nl = 64;
  const int ch = blockIdx.x * XAAC_LD_WAVES_PER_WG + wave;
  if (ch >= p.n_ch) return;
  int32_t *a = smem + wave * PER_WAVE, *b = a + 1024;
  /* the two window shapes, the F lines and (LD) the F / 2 old overlap words: every load the channel-frame starts with is in
     flight before the first is looked at (shapes, then lines, then -- behind the transform -- the overlap were three memory round
     trips in a kernel whose arithmetic takes less time than one) */
  const int32_t *spec = p.spec + (size_t)ch * F;
  int32_t *gov = p.overlap + (size_t)ch * NOV;
  const int shape_v = p.window_shape[ch], shape_prev_v = p.shape_prev[ch];
  constexpr int NSV = (F + 63) / 64, NOR = ELD ? 1 : (NOV + 63) / 64;
  constexpr int NE1 = ELD ? (F + 63) / 64 : 1, NE2 = ELD ? (2 * F + 63) / 64 : 1;
  int32_t sv[NSV], ovr[NOR], ov1[NE1], ov2[NE2];
#pragma unroll
  for (int k = 0; k < NSV; k++) sv[k] = lane + 64 * k < F ? spec[lane + 64 * k] : 0;
  if (!ELD) {
#pragma unroll
    for (int k = 0; k < NOR; k++) ovr[k] = lane + 64 * k < NOV ? gov[lane + 64 * k] : 0;
  } else { /* the 3 F old overlap words: see eld_overlap_add_regs */
#pragma unroll
    for (int k = 0; k < NE1; k++) ov1[k] = lane + 64 * k < F ? gov[lane + 64 * k] : 0;
#pragma unroll
    for (int k = 0; k < NE2; k++) ov2[k] = lane + 64 * k < 2 * F ? gov[F + lane + 64 * k] : 0;
  }
  const int shape = __builtin_amdgcn_readfirstlane(shape_v), shape_prev = __builtin_amdgcn_readfirstlane(shape_prev_v);
  if (shape > 1 || shape_prev > 1) { /* values the one-bit field cannot carry: left untouched */
    if (lane == 0 && p.status) p.status[ch] = XAAC_FATAL_BAD_WINDOW_SEQ;
    return;
  }
  int32_t acc = 0;
#pragma unroll
  for (int k = 0; k < NSV; k++) { /* the lines wait in the upper half of a: the pre twiddle is their only reader */
    if (lane + 64 * k < F) a[512 + lane + 64 * k] = sv[k];
    acc |= fx_abs_nrm(sv[k]);
  }
  const int e = fx_norm32(wave_or(acc)) - 1;
  x9_sync();
  const int q = xl_transform<F, ELD>(a + 512, a, b, e, lane, nl);
  int16_t *pcm = p.pcm16 + (size_t)(ch / p.ch_fac) * F * p.ch_fac + ch % p.ch_fac;
  if (ELD) {
    eld_overlap_add_regs<F>(a, ov1, ov2, gov, pcm, p.ch_fac, q, lane);
  } else { /* LD: the F / 2 old overlap words move into the free work array before the new ones overwrite their source */
#pragma unroll
    for (int k = 0; k < NOR; k++)
      if (lane + 64 * k < NOV) b[lane + 64 * k] = ovr[k];
    x9_sync();
    xl_ld_overlap_add<F>(a, b, gov, pcm, p.ch_fac, q, shape_prev, lane, nl);
  }
  if (lane == 0) {
    p.shape_prev[ch] = (uint8_t)shape; /* lpfuncs.c:800 */
    if (p.status) p.status[ch] = XAAC_OK;
  }
}

extern "C" hipError_t xaac_launch_imdct_ld(const xaac_imdct_ld_batch *p, hipStream_t stream) {
  const dim3 grid((p->n_ch + XAAC_LD_WAVES_PER_WG - 1) / XAAC_LD_WAVES_PER_WG), block(64 * XAAC_LD_WAVES_PER_WG);
  const size_t lds = XAAC_LD_LDS(p->frame_length, p->eld);
  if (p->frame_length == 512) {
    if (p->eld) hipLaunchKernelGGL((xaac_imdct_ld_kernel<512, true>), grid, block, lds, stream, *p);
    else hipLaunchKernelGGL((xaac_imdct_ld_kernel<512, false>), grid, block, lds, stream, *p);
  } else {
    if (p->eld) hipLaunchKernelGGL((xaac_imdct_ld_kernel<480, true>), grid, block, lds, stream, *p);
    else hipLaunchKernelGGL((xaac_imdct_ld_kernel<480, false>), grid, block, lds, stream, *p);
  }
  return hipGetLastError();
}

/* xaac_warm_up (xaac_abi.cpp): asking for a kernel's attributes puts this translation unit's code object on the device */
extern "C" hipError_t xaac_warm_imdct_ld(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&xaac_imdct_ld_kernel<512, false>));
}
