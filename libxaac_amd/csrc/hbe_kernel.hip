/*
 * hbe_kernel.hip -- the polyphase banks of the QMF-domain harmonic transposer on gfx950:
 *   xaac_hbe_synth_kernel  <-> ixheaacd_real_synth_filt    (decoder/ixheaacd_esbr_polyphase.c:157-274)
 *   xaac_hbe_anal_kernel   <-> ixheaacd_complex_anal_filt  (decoder/ixheaacd_esbr_polyphase.c:48-155)
 *   xaac_hbe_post_kernel   <-> ixheaacd_hbe_post_anal_process + the output stage of
 *                              ixheaacd_qmf_hbe_apply (decoder/ixheaacd_hbe_trans.c:1549-1571, :262-295)
 * The arithmetic is hbe_poly.h / hbe_trans.h (the oracle runs the same source sequentially).
 *
 * Mapping: one wave = one channel-frame.  The reference shifts a delay line per QMF column; here a column's transform
 * depends only on the column's input (hbe_poly.h), so the columns' transforms run side by side (lane = column, its
 * arrays in a private LDS strip with an odd stride), and the windowed sums -- 32 x synth_size outputs of ten products
 * each, 16 x 4 synth_size of five -- are spread over all lanes with coalesced stores.  The delay lines are rewritten
 * once per frame from the last columns instead of shifted per column.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hbe_trans.h"
#include "hbe_kernel.h"

namespace {
/* the frame's pitch and whether the channel takes part (inside the Path A chain both come from the side info / frame) */
__device__ __forceinline__ int hbe_pitch(const int32_t *pitch, const xaac_esbr_side *side, int ch) {
  return side ? side[ch].pitch_in_bins : (pitch ? pitch[ch] : 0);
}
__device__ __forceinline__ bool hbe_skip(const xaac_sbr_frame *frame, int ch) { return frame && !frame[ch].apply_processing; }
}  // namespace

__global__ __launch_bounds__(64) void xaac_hbe_synth_kernel(XaacHbeSynParams p) {
  extern __shared__ float lds[];
  float(*vv)[40] = reinterpret_cast<float(*)[40]>(lds); /* [9 + 32][2 s <= 40] */
  float *scr = lds + 41 * 40;
  const int ch = blockIdx.x, lane = threadIdx.x;
  xaac_hbe_state *st = p.state + ch;
  if (hbe_skip(p.frame, ch)) return;
  const int s = st->synth_size, ks = st->k_start, nc = p.num_columns;
  const bool bad = p.apply ? !xh_apply_params_ok(st, hbe_pitch(p.pitch, p.side, ch))
                           : (!xh_size_ok(s) || ks < 0 || ks + s > 64 || ks * 32 + 2 * s > 7 * 64 || nc < 0 || nc > 32);
  if (lane == 0 && p.status) p.status[ch] = bad ? -1 : 0;
  if (bad) return;
  /* apply mode, hbe_trans.c:235-248: the last synth_size samples of the previous frame's time signal move to the
     front; while the reference's FFT pointers are unset it re-initialises, which clears both delay lines */
  const bool cleared = p.apply && !st->fft_ready;
  float carry = 0.0f;
  if (p.apply && lane < s) carry = st->input_buf[nc * s + lane];
  if (cleared) {
    for (int e = lane; e < 640; e += 64) st->analy_buf[e] = 0.0f;
    for (int e = 20 * s + lane; e < 1280; e += 64) st->synth_buf[e] = 0.0f;
  }
  for (int e = lane; e < 9 * 2 * s; e += 64) {
    const int c = -1 - e / (2 * s), t = e % (2 * s);
    vv[c + 9][t] = cleared ? 0.0f : xh_synth_hist(st->synth_buf, s, c, t);
  }
  if (lane < nc)
    xh_synth_column(p.qmf_re + (size_t)ch * p.in_stride + lane * 64, p.qmf_im + (size_t)ch * p.in_stride + lane * 64, s, ks,
                    vv[lane + 9], scr + lane * 265);
  __syncthreads();
  const auto at = [&](int c, int t) { return vv[c + 9][t]; };
  for (int o = lane; o < nc * s; o += 64) st->input_buf[s + o] = xh_synth_out(at, s, o / s, o % s);
  if (p.apply && lane < s) st->input_buf[lane] = carry;
  for (int e = lane; e < 20 * s; e += 64) {
    const int c = nc - 1 - e / (2 * s);
    st->synth_buf[e] = vv[c + 9][e % (2 * s)]; /* nc >= 1: columns nc - 10 .. nc - 1 >= -9 */
  }
}

__global__ __launch_bounds__(64) void xaac_hbe_anal_kernel(XaacHbeAnaParams p) {
  extern __shared__ float lds[];
  float(*u)[80] = reinterpret_cast<float(*)[80]>(lds);            /* [16][2 a <= 80] */
  float(*res)[80] = reinterpret_cast<float(*)[80]>(lds + 16 * 80); /* [16][2 a <= 80] */
  float *scr = lds + 16 * 80 + 16 * 80;
  const int ch = blockIdx.x, lane = threadIdx.x;
  xaac_hbe_state *st = p.state + ch;
  if (hbe_skip(p.frame, ch)) return;
  const int s = st->synth_size, ks = st->k_start, a = 2 * s;
  const bool bad = p.apply ? !xh_apply_params_ok(st, hbe_pitch(p.pitch, p.side, ch)) : (!xh_size_ok(s) || ks < 0 || 4 * ks + 2 * a > 128);
  if (lane == 0 && p.status && !p.apply) p.status[ch] = bad ? -1 : 0;
  if (bad) return;
  constexpr int NCOL = XAAC_HBE_NO_BINS / 2;
  if (p.apply) /* hbe_trans.c:254-258: rows 16..27 become rows 0..11 (rows 12..27 are written below, after the barrier) */
    for (int e = lane; e < (XAAC_HBE_OPER_WIN_LEN - 1) * 128; e += 64) st->qmf_in_buf[e >> 7][e & 127] = st->qmf_in_buf[(e >> 7) + NCOL][e & 127];
  for (int e = lane; e < NCOL * 2 * a; e += 64) u[e / (2 * a)][e % (2 * a)] = xh_anal_u(st->input_buf, st->analy_buf, a, e / (2 * a), e % (2 * a));
  /* the delay line the last column leaves (read before anything of it is overwritten) */
  float nb[7];
#pragma unroll
  for (int q = 0; q < 7; q++) {
    const int n = lane + 64 * q;
    nb[q] = n < 10 * a ? xh_anal_x(st->input_buf, st->analy_buf, a, NCOL - 1, n) : 0.0f;
  }
  __syncthreads();
  if (lane < NCOL) xh_anal_column(u[lane], a, res[lane], scr + lane * 513);
  __syncthreads();
  for (int e = lane; e < NCOL * 128; e += 64) {
    const int idx = e >> 7, w = (e & 127) - 4 * ks;
    st->qmf_in_buf[idx + XAAC_HBE_OPER_WIN_LEN - 1][e & 127] = (w >= 0 && w < 2 * a) ? res[idx][w] : 0.0f;
  }
#pragma unroll
  for (int q = 0; q < 7; q++) {
    const int n = lane + 64 * q;
    if (n < 10 * a) st->analy_buf[n] = nb[q];
  }
}

/* 256 threads per channel.  Tiles of 16 output bands: thread (band, column) computes the column's block of products into
   LDS; then thread per complex element (row, band) moves the previous frame's upper rows down (rows 0..31 first, the
   rows they come from afterwards), adds the blocks that reach it in column order, and rotates rows 0..31 of the SBR
   range into the output (hbe_trans.c:262-295). */
__global__ __launch_bounds__(XAAC_HBE_POST_THREADS) void xaac_hbe_post_kernel(XaacHbePostParams p) {
  extern __shared__ float lds[];
  float(*blk)[XH_BLK] = reinterpret_cast<float(*)[XH_BLK]>(lds); /* [16 bands x 16 columns]: xh_column_block */
  const int ch = blockIdx.x, tid = threadIdx.x;
  xaac_hbe_state *st = p.state + ch;
  if (hbe_skip(p.frame, ch)) return;
  const int pitch = hbe_pitch(p.pitch, p.side, ch);
  if (!xh_apply_params_ok(st, pitch)) return;
  const int ms = st->max_stretch, sb0 = st->start_band, sb1 = st->end_band;
  int32_t xo[4];
  for (int q = 0; q < 4; q++) xo[q] = st->x_over_qmf[q];
  float *pv_re = p.pv_re + (size_t)ch * p.pv_stride, *pv_im = p.pv_im + (size_t)ch * p.pv_stride;
  const auto in = [&](int row, int band) {
    const float2 v = *reinterpret_cast<const float2 *>(&st->qmf_in_buf[row][2 * band]);
    const XhC c = {v.x, v.y};
    return c;
  };
  const float *in_flat = &st->qmf_in_buf[0][0];
  const auto inf = [&](int row, int idx) { return in_flat[128 * row + idx]; };
  for (int tile = 0; tile < 4; tile++) {
    {
      const int qb = 16 * tile + (tid >> 4), i = tid & 15;
      const int f = xh_band_factor(xo, ms, qb);
      if (f) xh_column_block(in, inf, f, qb, i, pitch, blk[tid]);
    }
    __syncthreads();
    for (int half = 0; half < 2; half++) {
      for (int e = tid; e < 32 * 16; e += XAAC_HBE_POST_THREADS) {
        const int r = 32 * half + (e >> 4), bt = e & 15, qb = 16 * tile + bt;
        float2 *dst = reinterpret_cast<float2 *>(&st->qmf_out_buf[r][2 * qb]);
        float2 v = make_float2(0.0f, 0.0f);
        if (!half) v = *reinterpret_cast<const float2 *>(&st->qmf_out_buf[r + 32][2 * qb]);
        const int f = xh_band_factor(xo, ms, qb);
        if (f) {
          const auto bk = [&](int i) { return (const float *)blk[16 * bt + i]; };
          v.x = xh_prod_gather(v.x, f, r, 0, bk);
          v.y = xh_prod_gather(v.y, f, r, 1, bk);
        }
        *dst = v;
        if (!half && qb >= sb0 && qb < sb1) {
          pv_re[64 * r + qb] = (float)(v.x * xaac_hbe_pv_cos[qb] - v.y * xaac_hbe_pv_sin[qb]);
          pv_im[64 * r + qb] = (float)(v.x * xaac_hbe_pv_sin[qb] + v.y * xaac_hbe_pv_cos[qb]);
        } else if (!half && p.zero_outside) {
          pv_re[64 * r + qb] = 0.0f;
          pv_im[64 * r + qb] = 0.0f;
        }
      }
      __syncthreads(); /* rows 32..63 are overwritten only after every row below has taken its start value from them */
    }
  }
  if (tid == 0 && !st->fft_ready && st->synth_size != 20) st->fft_ready = 1;
}

extern "C" hipError_t xaac_launch_hbe_synth(const XaacHbeSynParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_hbe_synth_kernel, dim3(p->n_ch), dim3(64), XAAC_HBE_SYN_LDS, stream, *p);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_hbe_anal(const XaacHbeAnaParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_hbe_anal_kernel, dim3(p->n_ch), dim3(64), XAAC_HBE_ANA_LDS, stream, *p);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_hbe_post(const XaacHbePostParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_hbe_post_kernel, dim3(p->n_ch), dim3(XAAC_HBE_POST_THREADS), XAAC_HBE_POST_LDS, stream, *p);
  return hipGetLastError();
}
