/*
 * imdct960_kernel.hip -- gfx950 kernel for the 960-line AAC IMDCT + windowing + overlap-add (frame_length 960 of
 * ixheaacd_imdct_process, decoder/ixheaacd_lpfuncs.c:347-802); stages and arithmetic in imdct960.h.
 *
 * Mapping: one wave = one channel-frame, four per workgroup (they share nothing, so only wave-level barriers).  The 960
 * lines and the 480 old overlap words are read once, coalesced, into LDS (the block exponent is an OR over the wave on
 * the way, the old overlap moves into the free work array after the transform); every stage of imdct960.h then spreads its
 * independent items over the 64 lanes between two 3.75 KB LDS arrays: pre twiddle in the prime-factor input order, 15 x 32-point (three passes of 120 items), 32 x 15-point as
 * 96 five-point + 160 three-point items, post twiddle with the 17476 scale, windowing / overlap-add straight to global
 * memory.  HBM traffic = 3.75 KB lines + 1.9 KB overlap in, 3.75 KB samples + 1.9 KB overlap out per channel-frame.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "imdct960.h"
#include "imdct960_kernel.h"

namespace {
__device__ __forceinline__ int32_t wave_or(int32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
  return v;
}
}  // namespace

__global__ __launch_bounds__(64 * XAAC_I960_WAVES_PER_WG) void xaac_imdct960_kernel(xaac_imdct_batch p) {
  extern __shared__ __attribute__((aligned(16))) int32_t smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nl = 64;
  const int ch = blockIdx.x * XAAC_I960_WAVES_PER_WG + wave;
  if (ch >= p.n_ch) return;
  int32_t *y = smem + wave * (960 + 960), *a = y + 960;
  /* the window words and the 960 lines: in flight together (the lines used to wait for the window check's round trip) */
  const int32_t *spec = p.spec + (size_t)ch * 960;
  int32_t *gov = p.overlap + (size_t)ch * 480;
  static_assert(sizeof(p.ics[0]) == 2 && sizeof(p.state[0]) == 2, "two bytes each: sequence, shape");
  const int ics_v = *reinterpret_cast<const uint16_t *>(p.ics + ch), st_v = *reinterpret_cast<const uint16_t *>(p.state + ch);
  int32_t sv[15];
#pragma unroll
  for (int k = 0; k < 15; k++) sv[k] = spec[lane + 64 * k];
  const int ics_bits = __builtin_amdgcn_readfirstlane(ics_v), st_bits = __builtin_amdgcn_readfirstlane(st_v);
  const int seq = ics_bits & 0xff, shape = ics_bits >> 8, pseq = st_bits & 0xff, pshape = st_bits >> 8;
  if (seq > 3 || shape > 1 || pseq > 3 || pshape > 1) { /* values the bitstream fields cannot carry: left untouched */
    if (lane == 0 && p.status) p.status[ch] = XAAC_FATAL_BAD_WINDOW_SEQ;
    return;
  }
  int32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 15; k++) {
    a[lane + 64 * k] = sv[k];
    acc |= fx_abs_nrm(sv[k]);
  }
  const int headroom = fx_norm32(wave_or(acc));
  x9_sync();
  const bool edge = pseq == X9_LONG_START || pseq == X9_EIGHT_SHORT;
  const size_t unit = (size_t)(ch / p.ch_fac) * 960 * p.ch_fac + ch % p.ch_fac;
  const X9Sink sk = {p.out32 ? p.out32 + unit : nullptr, p.pcm16 ? p.pcm16 + unit : nullptr, p.ch_fac, x9_qshift_adj(seq, edge),
                     p.pcm_mode};
  /* the lines sit in the work array: the pre twiddle is their only reader and ends before anything is written there */
  x9_imdct_process(a, gov, gov, y, a, headroom, seq, shape, pseq, pshape, sk, lane, nl, true);
  if (lane == 0) {
    p.state[ch].window_sequence = (uint8_t)seq; /* lpfuncs.c:800-801 */
    p.state[ch].window_shape = (uint8_t)shape;
    if (p.qshift_adj) p.qshift_adj[ch] = (int8_t)sk.qadj;
    if (p.status) p.status[ch] = XAAC_OK;
  }
}

extern "C" hipError_t xaac_launch_imdct960(const xaac_imdct_batch *p, hipStream_t stream) {
  const int wgs = (p->n_ch + XAAC_I960_WAVES_PER_WG - 1) / XAAC_I960_WAVES_PER_WG;
  hipLaunchKernelGGL(xaac_imdct960_kernel, dim3(wgs), dim3(64 * XAAC_I960_WAVES_PER_WG), XAAC_I960_LDS, stream, *p);
  return hipGetLastError();
}

/* xaac_warm_up (xaac_abi.cpp): asking for a kernel's attributes puts this translation unit's code object on the device */
extern "C" hipError_t xaac_warm_imdct960(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&xaac_imdct960_kernel));
}
