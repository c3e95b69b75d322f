/*
 * usac_imdct_kernel.hip -- gfx950 kernel for the USAC frequency-domain IMDCT + windowing + overlap-add of
 * ixheaacd_fd_frm_dec (decoder/ixheaacd_imdct.c:596: ccfl 1024 or 768; behind an FD or an LPD frame, with or without the FAC signal), arithmetic in usac_imdct.h.
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
#include "usac_fac.h"
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

struct LdsStrided { /* words of the sub-sequence T j + s of an interleaved (re, im) array */
  int32_t *p;
  int T, s;
  __device__ __forceinline__ int32_t &operator[](int w) const { return p[2 * (T * (w >> 1) + s) + (w & 1)]; }
};

/* L = ccfl (1024 or 768).  !SHORT: one transform of L lines (N = L/2 complex points); SHORT: eight of L/8 lines side by
   side (N = L/16).  For ccfl 768 a block's points go through three M-point transforms (M = 128 | 16) and the
   three-point stage (usac_imdct.h); every pass is spread over the lanes: L/8 radix-4 butterflies per pass either way. */
template <int L, bool SHORT>
__device__ __forceinline__ int transform(const int32_t *coef, const int2 *c, int32_t *A, int32_t *B, int lane, int shiftp,
                                         bool &all_zero) {
  constexpr int N = SHORT ? L / 16 : L / 2;   /* complex points per block */
  constexpr int M = xu_sub_points<N>();       /* ... of the power-of-two transform they go through */
  constexpr int T = N / M;                    /* 1, or 3 thirds */
  constexpr int PP = L / 128;                 /* line pairs (= points) per lane */
  int32_t xa[PP], xb[PP];
  /* Point q = lane + 64 m (block q / N, point i = q % N) needs the lines (x[2 i], x[2 N - 1 - 2 i]) of its block.  The frame's
     lines arrived in natural order, lane q's register m = (x[2 q], x[2 q + 1]) (frame(): one 8-byte load per lane and
     register, asked for before anything else of the channel-frame is known); the second line of point i is the odd line of
     pair N - 1 - i of the block: the mirrored lane's mirrored register in a long frame (N = 64 PP), the mirrored lane's
     same register in a short frame of 64-point blocks.  The 48-point blocks of a short 768-line frame do not fall on lane
     boundaries: those lines are read again (they are in the cache). */
#pragma unroll
  for (int m = 0; m < PP; m++) {
    xa[m] = c[m].x;
    if (!SHORT)
      xb[m] = __shfl(c[PP - 1 - m].y, 63 - lane);
    else if (N == 64)
      xb[m] = __shfl(c[m].y, 63 - lane);
    else {
      const int q = lane + 64 * m, blk = q / N, i = q % N;
      xb[m] = coef[2 * N * blk + 2 * N - 1 - 2 * i];
    }
  }
  int32_t mx = 0;
#pragma unroll
  for (int m = 0; m < PP; m++) {
    const int32_t a = fx_abs_sat(xa[m]), b = fx_abs_sat(xb[m]);
    mx = a > mx ? a : mx;
    mx = b > mx ? b : mx;
  }
  int s = fx_norm32(wave_max(mx));
  shiftp += s + 6 + xu_imdct_q_gain<N>();
#pragma unroll
  for (int m = 0; m < PP; m++) {
    const int q = lane + 64 * m, blk = q / N, i = q % N;
    int32_t a = fx_shlw(xa[m], s), b = fx_shlw(xb[m], s);
    if (T == 3) { /* ixheaacd_acelp_imdct's prescale of the 768- and 96-line blocks */
      a = xu_third_twice(a);
      b = xu_third_twice(b);
    }
    const XuCx v = xu_pre_twiddle<N>(a, b, i);
    *reinterpret_cast<int2 *>(A + 2 * N * blk + 2 * i) = make_int2(v.r, v.i);
  }
  wave_sync();
  constexpr int NBF = L / 8; /* radix-4 butterflies per pass over all blocks and thirds */
#pragma unroll
  for (int m = 0; m < 2; m++) {
    const int b = lane + 64 * m;
    if (b < NBF) {
      const int unit = b / (M / 4), bb = b % (M / 4), blk = unit / T, th = unit % T;
      const LdsStrided a = {A + 2 * N * blk, T, th};
      const Lds y = {B + 2 * N * blk + 2 * M * th};
      xu_fft_first<M>(a, y, bb);
    }
  }
  wave_sync();
#pragma unroll
  for (int del = 4; del < M / 2; del *= 4) {
#pragma unroll
    for (int m = 0; m < 2; m++) {
      const int b = lane + 64 * m;
      if (b < NBF) {
        const int unit = b / (M / 4), bb = b % (M / 4), blk = unit / T, th = unit % T;
        const Lds y = {B + 2 * N * blk + 2 * M * th};
        xu_fft_pass<M>(y, del, bb);
      }
    }
    wave_sync();
  }
  if (xu_not_pow4<M>()) { /* long frames only: one transform, or its three thirds */
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int b = lane + 64 * m;
      if (b < T * (M / 2)) {
        const Lds y = {B + 2 * M * (b / (M / 2))};
        xu_fft_last<M>(y, b % (M / 2));
      }
    }
    wave_sync();
  }
  /* (three-point stage;) post twiddle; second block exponent; the renormalised lines go to A in natural order */
  mx = 0;
  if (T == 1) {
#pragma unroll
    for (int m = 0; m < PP; m++) {
      const int q = lane + 64 * m, blk = q / N, i = q % N;
      const int2 t = *reinterpret_cast<const int2 *>(B + 2 * N * blk + 2 * i);
      const XuCx in = {t.x, t.y};
      const XuCx v = xu_post_twiddle<N>(in, i);
      xa[m] = v.r;
      xb[m] = v.i;
    }
  } else {
#pragma unroll
    for (int m = 0; m < PP / 3; m++) { /* a lane's groups: g of block blk -> points g, M + g, 2 M + g */
      const int q = lane + 64 * m, blk = q / M, g = q % M;
      const int32_t *y = B + 2 * N * blk;
      const int2 t0 = *reinterpret_cast<const int2 *>(y + 2 * g), t1 = *reinterpret_cast<const int2 *>(y + 2 * M + 2 * g),
                 t2 = *reinterpret_cast<const int2 *>(y + 4 * M + 2 * g);
      const XuCx x0 = {t0.x, t0.y}, x1 = {t1.x, t1.y}, x2 = {t2.x, t2.y};
      XuCx o[3];
      xu_p3_group<M>(x0, x1, x2, g, o);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const XuCx v = xu_post_twiddle<N>(o[k], k * M + g);
        xa[3 * m + k] = v.r;
        xb[3 * m + k] = v.i;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < PP; m++) {
    const int32_t a = fx_abs_sat(xa[m]), b = fx_abs_sat(xb[m]);
    mx = a > mx ? a : mx;
    mx = b > mx ? b : mx;
  }
  s = fx_norm32(wave_max(mx));
#pragma unroll
  for (int m = 0; m < PP; m++) {
    int blk, i;
    if (T == 1) {
      const int q = lane + 64 * m;
      blk = q / N;
      i = q % N;
    } else {
      const int q = lane + 64 * (m / 3);
      blk = q / M;
      i = (m % 3) * M + q % M;
    }
    A[2 * N * blk + 2 * i] = xu_normalize(xa[m], s - 1);
    A[2 * N * blk + 2 * N - 1 - 2 * i] = xu_normalize(xb[m], s - 1);
  }
  wave_sync();
  all_zero = s == 31; /* fx_norm32 of a block maximum of 0 */
  shiftp += s - 1;
  if (shiftp - XU_SHIFT_OLAP > 31) shiftp = 31 + XU_SHIFT_OLAP;
  return shiftp;
}

/* the old overlap word of the lane's current sample: every position of the windowing reads overlap[i] for output sample
   i and nothing else of it (usac_imdct.h: xu_long_sample_lpd / xu_short_sample_lpd), so the 16 words a lane needs are fetched
   with the lines, in front of the transform, instead of behind it */
struct OvReg {
  int32_t cur;
  __device__ __forceinline__ int32_t operator[](int) const { return cur; }
};

/* one channel-frame of L = ccfl lines; returns XAAC_OK or the status of a refused frame (nothing written then) */
template <int L>
__device__ __forceinline__ int frame(const XaacUsacImdctParams &p, int ch, int32_t *A, int32_t *B, int lane) {
  constexpr int SP = L / 64; /* samples per lane */
  constexpr int PP = L / 128;
  const int32_t *coef = p.coef + (size_t)ch * L;
  int32_t *gov = p.overlap + (size_t)ch * L;
  /* everything the channel-frame reads, asked for together: window words, LPD flags, FAC exponent, lines, old overlap (the
     window check, the FAC checks and the transform then start one memory latency after the wave does, not four) */
  const int seq = p.ics[ch].window_sequence, shape = p.ics[ch].window_shape, shape_prev = p.shape_prev[ch];
  const int flags = p.lpd_flags ? p.lpd_flags[ch] : 0;
  const int fac_q_in = p.fac ? p.fac[ch].q : 0;
  int2 c[PP];
  int32_t ovr[SP];
#pragma unroll
  for (int m = 0; m < PP; m++) c[m] = *reinterpret_cast<const int2 *>(coef + 2 * (lane + 64 * m));
#pragma unroll
  for (int m = 0; m < SP; m++) ovr[m] = gov[lane + 64 * m];
  if (seq > 4 || shape > 1 || shape_prev > 1) return XAAC_FATAL_BAD_WINDOW_SEQ; /* values the bitstream fields cannot carry */
  const XuLpd lp = {flags & 1, (flags >> 1) & 1, (p.fac && (flags & 2)) ? fac_q_in : 0};
  /* FAC data only ever follows an LPD frame and needs its signal; ccfl 1024 has no 256-tap KBD window (calc_window fails) */
  if (lp.fac && (!lp.td_prev || !p.fac)) return XAAC_FATAL_BAD_ARG;
  if (lp.fac && fac_q_in == XAAC_USAC_FAC_REFUSED) return XAAC_FATAL_BAD_ARG; /* xaac_usac_fac_kernel: ixheaacd_cal_fac_data's error */
  if (xu_lpd_window_missing<L>(lp.td_prev != 0, seq, shape_prev)) return XAAC_FATAL_BAD_WINDOW_SEQ;
  bool all_zero;
  const int shiftp = seq == 2 ? transform<L, true>(coef, c, A, B, lane, 0, all_zero) : transform<L, false>(coef, c, A, B, lane, 0, all_zero);
  if (lp.fac && (seq == 2 || seq == 3 || seq == 4) && !xu_fac_q_ok(shiftp, all_zero, seq == 2, lp.fac_q)) return XAAC_FATAL_BAD_ARG;
  const Lds x = {A};
  const Glb fac = {lp.fac ? p.fac[ch].data : gov};
  int32_t out[SP], nov[SP];
  if (seq != 2) {
    const bool stop_like = seq == 3 || seq == 4;
    const int oq = xu_long_output_q_lpd(shiftp, stop_like, lp);
#pragma unroll
    for (int m = 0; m < SP; m++) {
      const int i = lane + 64 * m;
      const OvReg ov = {ovr[m]};
      out[m] = xu_scale_adj(xu_long_sample_lpd<L>(x, ov, fac, i, shiftp, stop_like, shape_prev, lp), oq);
      nov[m] = xu_long_overlap<L>(x, i, shiftp);
    }
  } else {
    const int oq = xu_long_output_q(shiftp);
#pragma unroll
    for (int m = 0; m < SP; m++) {
      const int i = lane + 64 * m;
      const OvReg ov = {ovr[m]}; /* (positions L + i read no overlap) */
      out[m] = xu_scale(xu_short_sample_lpd<L>(x, ov, fac, i, shiftp, shape, shape_prev, lp), oq, 15);
      nov[m] = xu_scale(xu_short_sample_lpd<L>(x, ov, fac, L + i, shiftp, shape, shape_prev, lp), oq, XU_SHIFT_OLAP);
    }
  }
#pragma unroll
  for (int m = 0; m < SP; m++) {
    const int i = lane + 64 * m;
    if (lp.td_prev) out[m] = xu_float_round_trip(out[m]); /* imdct.c:459-470 around the LPD decoder's post filter */
    gov[i] = nov[m];
    if (p.out32) p.out32[(size_t)ch * L + i] = out[m];
    if (p.time) p.time[(size_t)ch * L + i] = (float)out[m] * 0.000030517578125f; /* ext_ch_ele.c:1008-1012 */
  }
  return XAAC_OK;
}

}  // namespace

#ifndef XU_MIN_WAVES
#define XU_MIN_WAVES 1
#endif
__global__ __launch_bounds__(64 * XAAC_USAC_WAVES_PER_WG, XU_MIN_WAVES) void xaac_usac_imdct_kernel(XaacUsacImdctParams p) {
  extern __shared__ __attribute__((aligned(16))) int32_t smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ch = blockIdx.x * XAAC_USAC_WAVES_PER_WG + wave;
  if (ch >= p.n_ch) return;
  int32_t *A = smem + wave * 2048, *B = A + 1024;
  const int rc = p.ccfl == 768 ? frame<768>(p, ch, A, B, lane) : frame<1024>(p, ch, A, B, lane);
  if (lane == 0) {
    if (rc == XAAC_OK) p.shape_prev[ch] = (uint8_t)p.ics[ch].window_shape; /* ext_ch_ele.c:1015 */
    if (p.status) p.status[ch] = rc;
  }
}

/* ixheaacd_cal_fac_data (imdct.c:210) for the channels whose frame follows an LPD frame and carries FAC data: one wave per
   channel, the work arrays in LDS, arithmetic and order of usac_fac.h -- element-wise loops over the lanes, the 24- to 64-point
   transform and the order-16 recursion on one lane (a frame in a few hundred of a switched stream's comes here) */
__global__ __launch_bounds__(64) void xaac_usac_fac_kernel(XaacUsacFacParams p) {
  __shared__ XfWork w;
  const int ch = blockIdx.x, lane = threadIdx.x;
  const int flags = p.lpd_flags[ch];
  if ((flags & 3) != 3) return; /* (uniform) */
  const int seq = p.ics[ch].window_sequence;
  const int lfac = seq == 2 ? p.ccfl >> 4 : p.ccfl >> 3; /* imdct.c:620-626, td_frame_prev set */
  const XfCx cx = {lane, 64};
  int32_t q = 0;
  const int rc = xf_cal_fac_data(cx, &w, p.in + ch, p.ccfl, lfac, p.out[ch].data, &q);
  if (lane == 0) p.out[ch].q = rc ? XAAC_USAC_FAC_REFUSED : w.s_q_out;
}

extern "C" hipError_t xaac_launch_usac_fac(const XaacUsacFacParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_usac_fac_kernel, dim3(p->n_ch), dim3(64), 0, stream, *p);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_usac_imdct(const XaacUsacImdctParams *p, hipStream_t stream) {
  const int wgs = (p->n_ch + XAAC_USAC_WAVES_PER_WG - 1) / XAAC_USAC_WAVES_PER_WG;
  hipLaunchKernelGGL(xaac_usac_imdct_kernel, dim3(wgs), dim3(64 * XAAC_USAC_WAVES_PER_WG), XAAC_USAC_LDS, stream, *p);
  return hipGetLastError();
}

/* xaac_warm_up (xaac_abi.cpp): asking for a kernel's attributes puts this translation unit's code object on the device */
extern "C" hipError_t xaac_warm_usac_imdct(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&xaac_usac_imdct_kernel));
}
