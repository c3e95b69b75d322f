/*
 * usac_imdct_kernel.hip -- gfx950 kernel for the USAC frequency-domain IMDCT + windowing + overlap-add of
 * ixheaacd_fd_frm_dec (decoder/ixheaacd_imdct.c:596: ccfl 1024, no FAC, previous frame FD), arithmetic in usac_imdct.h.
 *
 * Mapping: one wave = one channel-frame, four per workgroup (they share nothing).  The 1024 lines are read once, strided
 * so that lane i holds the pairs (x[2i], x[2N-1-2i]) its pre twiddle needs; the block exponent is a wave max.  The
 * transform is the reference's radix-4 decimation-in-time network (digit-reversed first pass, N/4 butterflies per pass,
 * a radix-2 pass for 512 points) between two 4 KB LDS arrays, every pass spread over the 64 lanes (the eight 64-point
 * transforms of a short frame run side by side: 8 x 16 butterflies per pass).  The post twiddle keeps its 16 results per
 * lane in registers through the second block-exponent reduction, the windowing / overlap-add then produces output sample
 * i and new-overlap sample i from closed forms (no intermediate 2048-word buffer), lanes = consecutive samples.
 * HBM traffic = 4 KB lines + 4 KB overlap in, 4 KB overlap + 4 KB output out per channel-frame.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "usac_imdct.h"
#include "usac_imdct_kernel.h"

namespace {

struct Lds {
  int32_t *p;
  __device__ __forceinline__ int32_t &operator[](int i) const { return p[i]; }
};
struct Glb {
  const int32_t *p;
  __device__ __forceinline__ int32_t operator[](int i) const { return p[i]; }
};

__device__ __forceinline__ int32_t wave_max(int32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int32_t t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

/* N = 512: one transform of 1024 lines; N = 64: eight of 128 lines side by side */
template <int N>
__device__ __forceinline__ int transform(const int32_t *coef, int32_t *A, int32_t *B, int lane, int shiftp) {
  constexpr int NB = 512 / N;                 /* blocks */
  int32_t xa[8], xb[8];
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int blk = N == 512 ? 0 : m, i = N == 512 ? lane + 64 * m : lane;
    xa[m] = coef[2 * N * blk + 2 * i];
    xb[m] = coef[2 * N * blk + 2 * N - 1 - 2 * i];
  }
  int32_t mx = 0;
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int32_t a = fx_abs_sat(xa[m]), b = fx_abs_sat(xb[m]);
    mx = a > mx ? a : mx;
    mx = b > mx ? b : mx;
  }
  int s = fx_norm32(wave_max(mx));
  shiftp += s + 6 + xu_imdct_q_gain<N>();
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int blk = N == 512 ? 0 : m, i = N == 512 ? lane + 64 * m : lane;
    const XuCx v = xu_pre_twiddle<N>(fx_shlw(xa[m], s), fx_shlw(xb[m], s), i);
    *reinterpret_cast<int2 *>(A + 2 * N * blk + 2 * i) = make_int2(v.r, v.i);
  }
  wave_sync();
#pragma unroll
  for (int m = 0; m < 2; m++) {
    const int b = lane + 64 * m, blk = N == 512 ? 0 : b / (N / 4), bb = N == 512 ? b : b % (N / 4);
    const Lds a = {A + 2 * N * blk}, y = {B + 2 * N * blk};
    xu_fft_first<N>(a, y, bb);
  }
  wave_sync();
#pragma unroll
  for (int del = 4; del < N / 2; del *= 4) {
#pragma unroll
    for (int m = 0; m < 2; m++) {
      const int b = lane + 64 * m, blk = N == 512 ? 0 : b / (N / 4), bb = N == 512 ? b : b % (N / 4);
      const Lds y = {B + 2 * N * blk};
      xu_fft_pass<N>(y, del, bb);
    }
    wave_sync();
  }
  if (N == 512) {
    const Lds y = {B};
#pragma unroll
    for (int m = 0; m < 4; m++) xu_fft_last512(y, lane + 64 * m);
    wave_sync();
  }
  /* post twiddle; second block exponent; the renormalised lines go to A in natural order */
  mx = 0;
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int blk = N == 512 ? 0 : m, i = N == 512 ? lane + 64 * m : lane;
    const int2 t = *reinterpret_cast<const int2 *>(B + 2 * N * blk + 2 * i);
    const XuCx in = {t.x, t.y};
    const XuCx v = xu_post_twiddle<N>(in, i);
    xa[m] = v.r;
    xb[m] = v.i;
    const int32_t a = fx_abs_sat(v.r), b = fx_abs_sat(v.i);
    mx = a > mx ? a : mx;
    mx = b > mx ? b : mx;
  }
  s = fx_norm32(wave_max(mx));
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const int blk = N == 512 ? 0 : m, i = N == 512 ? lane + 64 * m : lane;
    A[2 * N * blk + 2 * i] = xu_normalize(xa[m], s - 1);
    A[2 * N * blk + 2 * N - 1 - 2 * i] = xu_normalize(xb[m], s - 1);
  }
  wave_sync();
  shiftp += s - 1;
  if (shiftp - XU_SHIFT_OLAP > 31) shiftp = 31 + XU_SHIFT_OLAP;
  (void)NB;
  return shiftp;
}

}  // namespace

__global__ __launch_bounds__(64 * XAAC_USAC_WAVES_PER_WG) void xaac_usac_imdct_kernel(XaacUsacImdctParams p) {
  extern __shared__ __attribute__((aligned(16))) int32_t smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ch = blockIdx.x * XAAC_USAC_WAVES_PER_WG + wave;
  if (ch >= p.n_ch) return;
  int32_t *A = smem + wave * 2048, *B = A + 1024;
  const int seq = p.ics[ch].window_sequence, shape = p.ics[ch].window_shape, shape_prev = p.shape_prev[ch];
  if (seq > 4 || shape > 1 || shape_prev > 1) { /* values the bitstream fields cannot carry: left untouched */
    if (lane == 0 && p.status) p.status[ch] = XAAC_FATAL_BAD_WINDOW_SEQ;
    return;
  }
  const int32_t *coef = p.coef + (size_t)ch * 1024;
  int32_t *gov = p.overlap + (size_t)ch * 1024;
  const int shiftp = seq == 2 ? transform<64>(coef, A, B, lane, 0) : transform<512>(coef, A, B, lane, 0);
  const int oq = xu_long_output_q(shiftp);
  const Lds x = {A};
  const Glb ov = {gov};
  int32_t out[16], nov[16];
  if (seq != 2) {
    const bool stop_like = seq == 3 || seq == 4;
#pragma unroll
    for (int m = 0; m < 16; m++) {
      const int i = lane + 64 * m;
      out[m] = xu_scale_adj(xu_long_sample(x, ov, i, shiftp, stop_like, shape_prev), oq);
      nov[m] = xu_long_overlap(x, i, shiftp);
    }
  } else {
#pragma unroll
    for (int m = 0; m < 16; m++) {
      const int i = lane + 64 * m;
      out[m] = xu_scale(xu_short_sample(x, ov, i, shiftp, shape, shape_prev), oq, 15);
      nov[m] = xu_scale(xu_short_sample(x, ov, 1024 + i, shiftp, shape, shape_prev), oq, XU_SHIFT_OLAP);
    }
  }
#pragma unroll
  for (int m = 0; m < 16; m++) {
    const int i = lane + 64 * m;
    gov[i] = nov[m];
    if (p.out32) p.out32[(size_t)ch * 1024 + i] = out[m];
    if (p.time) p.time[(size_t)ch * 1024 + i] = (float)out[m] * 0.000030517578125f; /* ext_ch_ele.c:1008-1012 */
  }
  if (lane == 0) {
    p.shape_prev[ch] = (uint8_t)shape; /* ext_ch_ele.c:1015 */
    if (p.status) p.status[ch] = XAAC_OK;
  }
}

extern "C" hipError_t xaac_launch_usac_imdct(const XaacUsacImdctParams *p, hipStream_t stream) {
  const int wgs = (p->n_ch + XAAC_USAC_WAVES_PER_WG - 1) / XAAC_USAC_WAVES_PER_WG;
  hipLaunchKernelGGL(xaac_usac_imdct_kernel, dim3(wgs), dim3(64 * XAAC_USAC_WAVES_PER_WG), XAAC_USAC_LDS, stream, *p);
  return hipGetLastError();
}
