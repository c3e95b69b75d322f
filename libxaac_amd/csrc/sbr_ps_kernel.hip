/*
 * sbr_ps_kernel.hip -- the parametric-stereo tool of HE-AACv2 on gfx950: what the reference runs inside the
 * left channel's synthesis loop (decoder/ixheaacd_qmf_dec.c:1015-1031 with ixheaacd_thumb_ps_dec.c:69 and
 * ixheaacd_ps_dec.c), pulled out in front of the two synthesis banks.
 *
 * Mapping: ONE WAVE = ONE STREAM.  The tool is a recursion over the 32 QMF slots of the frame (delay lines,
 * transient detector, interpolated mixing matrix), so slots are walked in order; inside a slot the lanes are
 * the hybrid sub-bands / QMF bands / transient bins / parameter groups (sbr_ps.h, the source the oracle runs
 * sequentially).  The PS state (5.2 KB), the frame's side info and one slot of left/right samples live in LDS;
 * the QMF matrix stays where the core kernel put it (L2-resident workspace), one 512-byte row in and two out
 * per slot.  The kernel also does what ixheaacd_cplx_synt_qmffilt does around the tool -- bring the matrix to
 * the PS scale (adjust_scale, qmf_dec.c:937), the hybrid look-ahead's scale (thumb:77), the common shift in
 * front of the left bank (generic:1610) -- so that both synthesis launches find their input ready.
 */
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#ifdef XS_PROFILE
/* phase timers (tools/prof_sbr_core.py ps): cycles of lane 0 between XP_T hooks, summed over streams */
__shared__ long long xp_prof_last;
__shared__ long long xp_prof_acc[16];
#define XP_T(i)                                   \
  do {                                            \
    if (threadIdx.x == 0) {                       \
      long long t_ = clock64();                   \
      xp_prof_acc[i] += t_ - xp_prof_last;        \
      xp_prof_last = t_;                          \
    }                                             \
  } while (0)
#else
#define XP_T(i)
#endif
#include "sbr_ps.h"
#include "sbr_ps_kernel.h"

namespace {

struct XpLdsState {
  XAAC_PS_STATE_HEAD_FIELDS
};
constexpr int kHeadWords = offsetof(xaac_ps_state, syn_ring_r) / 4;
static_assert(offsetof(xaac_ps_state, syn_ring_r) % 4 == 0 && sizeof(XpLdsState) == kHeadWords * 4, "mirror layout");
static_assert(sizeof(xaac_ps_frame) % 4 == 0, "word copies");

struct XpLds {
  XpLdsState ps;
  xaac_ps_frame pf;
  XpHyb hy;
  int32_t left[128], right[128];
  int32_t hyb_u[3][2][44];  /* hybrid filter input of QMF bands 0..2: 12 slots of history + this frame's 32 */
  int32_t hyb_all[32][20];  /* hybrid sub-band samples of all 32 slots: re of sub-bands 0..9, then im */
  int16_t ratio[24];
  int32_t band_pw[64];
  XpTables tabs; /* the PS constants: a table lookup in global memory costs a slot-loop iteration its latency */
};
static_assert(sizeof(XpTables) % 4 == 0, "word copies");

/* global -> LDS with eight loads in flight (see sbr_core_kernel.hip) */
__device__ __forceinline__ void copy_words(int32_t *dst, const int32_t *src, int n, int lane) {
  for (int i = lane; i < n; i += 64 * 8) {
    int32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (i + 64 * j < n) t[j] = src[i + 64 * j];
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (i + 64 * j < n) dst[i + 64 * j] = t[j];
  }
}

__device__ __forceinline__ int32_t adj_word(int32_t v, int shift) { /* env_calc.c:1099 on one word */
  if (shift == 0) return v;
  if (shift > 31) shift = 31;
  if (shift < -31) shift = -31;
  return shift > 0 ? fx_shlw(v, shift) : (v >> -shift);
}

}  // namespace

__global__ __launch_bounds__(64) void xaac_ps_kernel(XaacPsParams p) {
  __shared__ XpLds s;
  const int n = blockIdx.x, lane = threadIdx.x;
  if (!(p.sbr_frame[n].apply_processing && p.header[n].channel_mode == 3)) { /* sbr_dec.c:1246: mono this frame */
    if (lane == 0) {
      p.par_l[8 * (size_t)n + 6] = 0;
      p.par_r[8 * (size_t)n + 6] = 1;
    }
    return;
  }
  xaac_ps_state *gps = p.state + n;
  int32_t *gx = p.x + (size_t)n * (40 * 128) + 2 * 128; /* slot 0 */
  int32_t *gr = p.xr + (size_t)n * (32 * 128);
  copy_words(reinterpret_cast<int32_t *>(&s.ps), reinterpret_cast<const int32_t *>(gps), kHeadWords, lane);
  copy_words(reinterpret_cast<int32_t *>(&s.pf), reinterpret_cast<const int32_t *>(p.frame + n),
             sizeof(xaac_ps_frame) / 4, lane);
  copy_words(reinterpret_cast<int32_t *>(&s.tabs), reinterpret_cast<const int32_t *>(&xaac_ps_tables),
             sizeof(XpTables) / 4, lane);
  __syncthreads();
#ifdef XS_PROFILE
  if (lane == 0) {
    for (int i = 0; i < 16; i++) xp_prof_acc[i] = 0;
    xp_prof_last = clock64();
  }
#endif
  const XsCx cx = {lane, 64};
  const int ps_clamped = xp_frame_sanitize(cx, &s.pf); /* indices a parser cannot produce: contained, reported */
  int16_t *par = p.par_l + 8 * (size_t)n;
  const int lb_scale = par[0], ov_lb_scale = par[1], hb_scale = par[2], st_syn = par[3], lsb = par[4], usb = par[5];
  const int ps_scale = xp_init_ps_scale(cx, &s.ps, lb_scale, ov_lb_scale, hb_scale); /* sbr_dec.c:1252 */
  const int ov_lb_shift = ps_scale - ov_lb_scale, lb_shift = ps_scale - lb_scale, hb_shift = ps_scale - hb_scale;
  const int common_shift = (st_syn - ps_scale) - 8;
  /* what adjust_scale would do to this lane's band in slots < 6 / >= 6 (qmf_dec.c:937-953) */
  const int sh_ov = lane < lsb ? ov_lb_shift : (lane < usb ? hb_shift : 0);
  const int sh_lb = lane < lsb ? lb_shift : (lane < usb ? hb_shift : 0);
  int env = 0;
  XP_T(1);
  /* ---- hybrid analysis of the whole frame (ixheaacd_hybrid_analysis, hybrid.c:214, is a 13-tap FIR on QMF bands
     0..2 looking six slots ahead: no recursion, so all 32 slots are filtered at once, one slot per lane, instead of
     three lanes per slot inside the slot loop).  Input of step l: row l + 6 as adjust_scale leaves it (slots of the
     next frame are not rescaled), then the delay-buffer shift of thumb_ps_dec.c:77. */
  if (lane < 32) {
    const int shiftdelay = lane < 32 - 6 ? 0 : (int16_t)(lb_scale - ps_scale);
#pragma unroll
    for (int b = 0; b < 3; b++) {
      const int sha = lane + 6 < 32 ? (b < lsb ? lb_shift : (b < usb ? hb_shift : 0)) : 0;
#pragma unroll
      for (int c = 0; c < 2; c++) {
        int32_t v = adj_word(gx[(lane + 6) * 128 + 64 * c + b], sha);
        v = shiftdelay < 0 ? fx_shl(v, -shiftdelay) : fx_shr(v, shiftdelay);
        s.hyb_u[b][c][12 + lane] = v;
      }
    }
  } else if (lane < 32 + 12) {
#pragma unroll
    for (int b = 0; b < 3; b++) {
      s.hyb_u[b][0][lane - 32] = s.ps.hyb_buf[b][0][lane - 32];
      s.hyb_u[b][1][lane - 32] = s.ps.hyb_buf[b][1][lane - 32];
    }
  }
  __syncthreads();
  if (lane < 32) { /* QMF band 0: eight-channel filter, six sub-bands */
    int32_t re[8], im[8];
    xp_filt_8ch(&s.tabs, &s.hyb_u[0][0][lane], &s.hyb_u[0][1][lane], re, im);
#pragma unroll
    for (int k = 0; k < 6; k++) {
      s.hyb_all[lane][k] = re[k];
      s.hyb_all[lane][10 + k] = im[k];
    }
  }
  { /* QMF bands 1 and 2: two sub-bands each */
    const int b = 1 + (lane >> 5), l = lane & 31;
    int32_t re[2], im[2];
    xp_filt_2ch(&s.tabs, &s.hyb_u[b][0][l], &s.hyb_u[b][1][l], re, im);
    s.hyb_all[l][4 + 2 * b] = re[0];
    s.hyb_all[l][5 + 2 * b] = re[1];
    s.hyb_all[l][14 + 2 * b] = im[0];
    s.hyb_all[l][15 + 2 * b] = im[1];
  }
  if (lane < 12) {
#pragma unroll
    for (int b = 0; b < 3; b++) {
      s.ps.hyb_buf[b][0][lane] = s.hyb_u[b][0][32 + lane];
      s.ps.hyb_buf[b][1][lane] = s.hyb_u[b][1][32 + lane];
    }
  }
  __syncthreads();
  XP_T(4);
  /* the next slot's row is fetched while the current slot is processed */
  int32_t n_re = gx[lane], n_im = gx[64 + lane];
  for (int l = 0; l < 32; l++) {
    {
      const int sh = l < 6 ? sh_ov : sh_lb;
      s.left[lane] = adj_word(n_re, sh);
      s.left[64 + lane] = adj_word(n_im, sh);
      if (lane < 10) {
        s.hy.l_re[lane] = s.hyb_all[l][lane];
        s.hy.l_im[lane] = s.hyb_all[l][10 + lane];
      }
      if (l + 1 < 32) {
        n_re = gx[(l + 1) * 128 + lane];
        n_im = gx[(l + 1) * 128 + 64 + lane];
      }
    }
    __syncthreads();
    XP_T(2);
    if (env <= XAAC_PS_MAX_ENV && l == s.pf.border_position[env]) {
      xp_init_rot_env(cx, &s.tabs, &s.ps, &s.pf, env, usb);
      env++;
    }
    XP_T(3);
    xp_decorrelation(cx, &s.tabs, &s.ps, &s.hy, s.left, s.right, s.ratio, s.band_pw);
    XP_T(5);
    xp_apply_rot(cx, &s.tabs, &s.ps, &s.hy, s.left, s.right);
    XP_T(6);
    for (int k = lane; k < 128; k += 64) {
      int32_t v = s.left[k];
      if (common_shift < 0)
        v = fx_shr(v, -common_shift > 31 ? 31 : -common_shift);
      else if (common_shift > 0)
        v = fx_shl_sat(v, common_shift);
      gx[l * 128 + k] = v;
      gr[l * 128 + k] = s.right[k];
    }
    __syncthreads();
    XP_T(7);
  }
  /* ---- state and the two synthesis launches' parameters ---- */
  {
    int32_t *dst = reinterpret_cast<int32_t *>(gps);
    const int32_t *src = reinterpret_cast<const int32_t *>(&s.ps);
    for (int i = lane; i < kHeadWords; i += 64) dst[i] = src[i];
  }
#ifdef XS_PROFILE
  XP_T(8);
  __syncthreads();
  if (lane < 16 && p.dbg) atomicAdd(reinterpret_cast<unsigned long long *>(p.dbg) + 32 + lane, (unsigned long long)xp_prof_acc[lane]);
#endif
  if (lane == 0) {
    int16_t *pr = p.par_r + 8 * (size_t)n;
    const int16_t ready = (int16_t)(st_syn - 8); /* makes the bank's own rescale a no-op: data is in place */
    par[0] = par[1] = par[2] = ready;
    pr[0] = pr[1] = pr[2] = (int16_t)ps_scale;
    pr[3] = gps->st_syn_scale_r;
    pr[4] = gps->syn_lsb_r;
    pr[5] = gps->syn_usb_r;
    pr[6] = 0;
    par[6] = 0;
    gps->lb_scale_r = gps->ov_lb_scale_r = gps->hb_scale_r = (int16_t)ps_scale; /* sbr_dec.c:1261-1264 */
    p.sbr_state[n].ps_scale = (int16_t)ps_scale;
#ifndef XS_PROFILE
    if (ps_clamped && p.status) p.status[n] = -1;
#endif
  }
}

extern "C" hipError_t xaac_launch_ps(const XaacPsParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_ps_kernel, dim3(p->n), dim3(64), 0, stream, *p);
  return hipGetLastError();
}
