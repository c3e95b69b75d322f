/*
 * esbr_core_kernel.hip -- gfx950 kernel for the float middle of the reference's default SBR path (Path A) on HE-AAC
 * channels: ixheaacd_generate_hf (sbrdec_lpfuncs.c:981), ixheaacd_sbr_env_calc (esbr_envcal.c:71), the regrouping of
 * ixheaacd_esbr_synthesis_regrp (sbr_dec.c:297) and the history shifts of sbr_dec.c:835-857; arithmetic in esbr_core.h.
 *
 * Mapping: one wave = one channel-frame, lane = QMF band.  The path works 38 slots behind the analysis bank (op_delay 6 +
 * the 32 slots the reference reserves for its harmonic transposer), so the HF generator reads the channel's 40-row
 * history straight from its state; this frame's analysis rows only enter the history at the end.  sbr_qmf_out lives in a
 * 42-row global scratch (L2-resident while the wave works on it), the per-band vectors in LDS: the stages are chains of
 * short dependent loops, so what the kernel needs is many resident waves -- with the matrix in LDS (21.5 KB, 6 waves per
 * CU) the same code measured 1.35x slower than with 4 KB of LDS and 24 waves per CU reading the matrix through L2.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "esbr_core.h"
#include "esbr_core_kernel.h"

__global__ __launch_bounds__(64) void xaac_esbr_core_kernel(XaacEsbrCoreParams p) {
  __shared__ XeWork w;
  const int ch = blockIdx.x, lane = threadIdx.x;
  const XsCx cx = {lane, 64};
  const xaac_sbr_header *h = p.header + ch;
  const xaac_sbr_frame *f = p.frame + ch;
  const xaac_esbr_side *sd = p.side + ch;
  xaac_esbr_state *st = p.state + ch;
  float *ore = p.out_re + (size_t)ch * XAAC_ESBR_OUT_ROWS * 64, *oim = p.out_im + (size_t)ch * XAAC_ESBR_OUT_ROWS * 64;
  float *rre = p.syn_re + (size_t)ch * XAAC_ESBR_L_ROWS * 64, *rim = p.syn_im + (size_t)ch * XAAC_ESBR_L_ROWS * 64;
  const float *are = p.ana_re + (size_t)ch * 2048, *aim = p.ana_im + (size_t)ch * 2048;
  const int apply = f->apply_processing != 0;
  int rc = 0;
  if (apply && xe_side_info_bad(h, f, sd)) rc = -1;
  /* sbr_qmf_out: 8 rows of history, the rest cleared (the stages write every cell that is read later) */
  for (int i = lane; i < XAAC_ESBR_OUT_ROWS * 64; i += 64) {
    const bool hist = apply && i < XAAC_ESBR_OUT_HIST_ROWS * 64;
    ore[i] = hist ? (&st->out_re[0][0])[i] : 0.0f;
    oim[i] = hist ? (&st->out_im[0][0])[i] : 0.0f;
  }
  __syncthreads();
  if (sd->qmf_sb_prev >= 0 && sd->qmf_sb_prev <= 64) xe_hbe_history_clear(cx, st, sd->qmf_sb_prev);
  const XeMat src = {&st->qmf_re[0][0] + 128, &st->qmf_im[0][0] + 128}, dst = {ore + 128, oim + 128};
  if (apply && rc == 0) {
    xe_generate_hf(cx, h, f, sd, st, &w, src, dst);
    __syncthreads();
    rc = w.err ? -1 : xe_env_calc(cx, h, f, sd, st, &w, dst, src);
  }
  __syncthreads();
  {
    const int stop = apply ? 2 * f->border_vec[0] : 0;
    for (int i = 0; i < 32; i++) { /* regrouping, sbr_dec.c:365-395 */
      const int xo = i < stop ? sd->qmf_sb_prev : h->sub_band_start;
      rre[64 * i + lane] = lane < xo ? st->qmf_re[2 + i][lane] : ore[64 * (2 + i) + lane];
      rim[64 * i + lane] = lane < xo ? st->qmf_im[2 + i][lane] : oim[64 * (2 + i) + lane];
    }
  }
  if (p.with_ps) /* the six look-ahead rows of the PS hybrid filter: bands 0..4 of qmf_buf rows 34..39 (sbr_dec.c:487-505) */
    for (int i = 32; i < 38; i++) {
      rre[64 * i + lane] = lane < 5 ? st->qmf_re[2 + i][lane] : 0.0f;
      rim[64 * i + lane] = lane < 5 ? st->qmf_im[2 + i][lane] : 0.0f;
    }
  __syncthreads();
  /* histories: rows 32.. of this frame's buffers become rows 0.. of the next frame's (sbr_dec.c:835-857) */
  {
    float t0[8], t1[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      t0[r] = st->qmf_re[32 + r][lane];
      t1[r] = st->qmf_im[32 + r][lane];
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      st->qmf_re[r][lane] = t0[r];
      st->qmf_im[r][lane] = t1[r];
    }
    for (int r = 0; r < 32; r++) {
      st->qmf_re[8 + r][lane] = lane < 32 ? are[64 * r + lane] : 0.0f;
      st->qmf_im[8 + r][lane] = lane < 32 ? aim[64 * r + lane] : 0.0f;
    }
    for (int r = 0; r < 8; r++) {
      st->out_re[r][lane] = ore[64 * (32 + r) + lane];
      st->out_im[r][lane] = oim[64 * (32 + r) + lane];
    }
  }
  if (lane == 0 && p.status) p.status[ch] = rc;
}

extern "C" hipError_t xaac_launch_esbr_core(const XaacEsbrCoreParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_esbr_core_kernel, dim3(p->n_ch), dim3(64), 0, stream, *p);
  return hipGetLastError();
}
