/*
 * hbe_kernel.hip -- the polyphase banks of the QMF-domain harmonic transposer on gfx950:
 *   xaac_hbe_synth_kernel  <-> ixheaacd_real_synth_filt    (decoder/ixheaacd_esbr_polyphase.c:157-274)
 *   xaac_hbe_anal_kernel   <-> ixheaacd_complex_anal_filt  (decoder/ixheaacd_esbr_polyphase.c:48-155)
 * The arithmetic is hbe_poly.h (the oracle runs the same source sequentially).
 *
 * Mapping: one wave = one channel-frame.  The reference shifts a delay line per QMF column; here a column's transform
 * depends only on the column's input (hbe_poly.h), so the columns' transforms run side by side (lane = column, its
 * arrays in a private LDS strip with an odd stride), and the windowed sums -- 32 x synth_size outputs of ten products
 * each, 16 x 4 synth_size of five -- are spread over all lanes with coalesced stores.  The delay lines are rewritten
 * once per frame from the last columns instead of shifted per column.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hbe_poly.h"
#include "hbe_kernel.h"

__global__ __launch_bounds__(64) void xaac_hbe_synth_kernel(XaacHbeSynParams p) {
  extern __shared__ float lds[];
  float(*vv)[40] = reinterpret_cast<float(*)[40]>(lds); /* [9 + 32][2 s <= 40] */
  float *scr = lds + 41 * 40;
  const int ch = blockIdx.x, lane = threadIdx.x;
  xaac_hbe_state *st = p.state + ch;
  const int s = st->synth_size, ks = st->k_start, nc = p.num_columns;
  const bool bad = !xh_size_ok(s) || ks < 0 || ks + s > 64 || ks * 32 + 2 * s > 7 * 64 || nc < 0 || nc > 32;
  if (lane == 0 && p.status) p.status[ch] = bad ? -1 : 0;
  if (bad) return;
  for (int e = lane; e < 9 * 2 * s; e += 64) {
    const int c = -1 - e / (2 * s), t = e % (2 * s);
    vv[c + 9][t] = xh_synth_hist(st->synth_buf, s, c, t);
  }
  if (lane < nc)
    xh_synth_column(p.qmf_re + ((size_t)ch * nc + lane) * 64, p.qmf_im + ((size_t)ch * nc + lane) * 64, s, ks, vv[lane + 9],
                    scr + lane * 265);
  __syncthreads();
  const auto at = [&](int c, int t) { return vv[c + 9][t]; };
  for (int o = lane; o < nc * s; o += 64) st->input_buf[s + o] = xh_synth_out(at, s, o / s, o % s);
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
  const int s = st->synth_size, ks = st->k_start, a = 2 * s;
  const bool bad = !xh_size_ok(s) || ks < 0 || 4 * ks + 2 * a > 128;
  if (lane == 0 && p.status) p.status[ch] = bad ? -1 : 0;
  if (bad) return;
  constexpr int NCOL = XAAC_HBE_NO_BINS / 2;
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

extern "C" hipError_t xaac_launch_hbe_synth(const XaacHbeSynParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_hbe_synth_kernel, dim3(p->n_ch), dim3(64), XAAC_HBE_SYN_LDS, stream, *p);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_hbe_anal(const XaacHbeAnaParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_hbe_anal_kernel, dim3(p->n_ch), dim3(64), XAAC_HBE_ANA_LDS, stream, *p);
  return hipGetLastError();
}
