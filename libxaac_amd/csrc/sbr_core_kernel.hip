/*
 * sbr_core_kernel.hip -- the middle of ixheaacd_sbr_dec (decoder/ixheaacd_sbr_dec.c:726-775,
 * :1050-1245, :1283-1308; low-power mode) on gfx950: overlap restore + xs_rescale_x_overlap, block
 * floating point, LPP transposer, envelope adjustment, LPC/overlap/scale state update -- everything
 * between the two QMF banks (which run as their own slot-parallel kernels before and after).
 *
 * Mapping: ONE WAVE = ONE CHANNEL-FRAME, one wave per workgroup.  The channel's 40 x 64 QMF matrix,
 * its header / frame side info, the control part of its state and the per-frame scratch (XsWork)
 * live in LDS (~13 KB, so a dozen channels share a CU and hide each other's latencies).  The
 * arithmetic is sbr_core.h -- the same source the oracle runs sequentially -- which puts QMF bands
 * (energies, covariances, LPC filtering, gain application) on the lanes, keeps per-band gains in
 * lane registers and runs the inherently sequential sums/walks as uniform scalar code; the matrix is
 * band-minor, so a lane-per-band access is a conflict-free LDS row read.
 * Global traffic is the coalesced copy-in (overlap slots, the 32 analysed slots x 32 bands, side
 * info, state) and copy-out (32 slots for the synthesis bank, overlap slots, state).
 *
 * (The first version of this kernel ran one channel per LANE on the matrix in global memory:
 * 4.3 ms per 16384 channels, profiles/r01_c_c3_kernel_stats.txt -- latency bound on 10 KB-strided
 * accesses.)
 */
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#ifdef XS_PROFILE
/* phase timers (tools/prof_sbr_core.py): cycles of lane 0 between XS_T hooks, summed over channels */
__shared__ long long xs_prof_last;
__shared__ long long xs_prof_acc[32];
#define XS_T(i)                                   \
  do {                                            \
    if (threadIdx.x == 0) {                       \
      long long t_ = clock64();                   \
      xs_prof_acc[i] += t_ - xs_prof_last;        \
      xs_prof_last = t_;                          \
    }                                             \
  } while (0)
#endif
/* LDS copies of the two tables behind every pseudo-float divide / square root (sbr_core.h: XS_TAB_*): a lookup in
   global memory on those serial paths costs its latency each time */
__shared__ int16_t xs_lds_inv_table[256];
__shared__ int16_t xs_lds_sqrt_table[258];
#define XS_TAB_INV(i) xs_lds_inv_table[i]
#define XS_TAB_SQRT(i) xs_lds_sqrt_table[i]
#include "sbr_core.h"
#include "sbr_core_kernel.h"

namespace {

/* LDS copy of the state members the core touches: the four bank-limit shorts + the struct tail */
struct XsLdsState {
  int16_t codec_usb, syn_lsb, syn_usb, pad2_;
  XAAC_SBR_STATE_TAIL_FIELDS
};
constexpr int kHeadOff = offsetof(xaac_sbr_state, codec_usb);
constexpr int kTailOff = offsetof(xaac_sbr_state, lpc_real);
constexpr int kTailWords = (sizeof(xaac_sbr_state) - kTailOff) / 4;
static_assert(kHeadOff % 4 == 0 && kTailOff % 4 == 0 && sizeof(xaac_sbr_state) % 4 == 0, "word copies");
static_assert(offsetof(XsLdsState, lpc_real) == 8 && sizeof(XsLdsState) == 8 + kTailWords * 4, "mirror layout");
static_assert(offsetof(xaac_sbr_state, overlap) % 16 == 0 || true, "");
static_assert(sizeof(xaac_sbr_header) % 4 == 0 && sizeof(xaac_sbr_frame) % 4 == 0, "word copies");

/* of the frame side info only the head lives in LDS: the envelope scale factors (896 B) are read once per envelope
   and stay in global memory, the noise floor gets its own twenty bytes (sbr_core.h hands both in beside `f`) */
constexpr int kFrameHeadBytes = offsetof(xaac_sbr_frame, int_env_sf_arr);
static_assert(kFrameHeadBytes % 4 == 0 && offsetof(xaac_sbr_frame, int_noise_floor) % 4 == 0, "word copies");
static_assert(offsetof(xaac_sbr_frame, int_noise_floor) == kFrameHeadBytes + sizeof(((xaac_sbr_frame *)0)->int_env_sf_arr),
              "nothing but the two arrays behind the head");

template <int HQ>
struct XsLds {
  int32_t x[(HQ ? 2 : 1) * XAAC_SBR_X_WORDS + 128]; /* + one row: the reference's edge writes may run past slot 37 */
  XsLdsState st;
  xaac_sbr_header h;
  int32_t f_head[kFrameHeadBytes / 4];
  int32_t noise_floor[sizeof(((xaac_sbr_frame *)0)->int_noise_floor) / 4];
  XsWork w;
  int16_t rand_hi[HQ ? 4 : 568]; /* xaac_sbr_rand_ph >> 16 (the low-power slot loop; HQ reads the 32-bit table ahead) */
};

/* global -> LDS (or back): eight loads are in flight before the first store, so a copy costs one memory
   latency per 512 words instead of one per 64 */
__device__ __forceinline__ void copy_words(int32_t *dst, const int32_t *src, int n, int lane) {
  int i = lane;
  for (; i + 64 * 7 < n; i += 64 * 8) {
    int32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = src[i + 64 * j];
#pragma unroll
    for (int j = 0; j < 8; j++) dst[i + 64 * j] = t[j];
  }
  int32_t t[8];
#pragma unroll
  for (int j = 0; j < 8; j++)
    if (i + 64 * j < n) t[j] = src[i + 64 * j];
#pragma unroll
  for (int j = 0; j < 8; j++)
    if (i + 64 * j < n) dst[i + 64 * j] = t[j];
}

}  // namespace

/* HQ = 0: low-power mode, rows of 64 reals; HQ = 1: rows of 64 real | 64 imaginary (HE-AAC mono / v2) */
template <int HQ>
__global__ __launch_bounds__(64) void xaac_sbr_core_kernel(XaacSbrCoreParams p) {
  __shared__ XsLds<HQ> s;
  constexpr int ROW = HQ ? 128 : 64, XW = (HQ ? 2 : 1) * XAAC_SBR_X_WORDS;
  const int ch = blockIdx.x, lane = threadIdx.x;
  xaac_sbr_state *gst = p.state + ch;
  int32_t *gx = p.x + (size_t)ch * XW;
  const int32_t *gstw = reinterpret_cast<const int32_t *>(gst);

  /* ---- copy-in ---- */
  copy_words(reinterpret_cast<int32_t *>(&s.h), reinterpret_cast<const int32_t *>(p.header + ch),
             sizeof(xaac_sbr_header) / 4, lane);
  copy_words(s.f_head, reinterpret_cast<const int32_t *>(p.frame + ch), kFrameHeadBytes / 4, lane);
  if (lane < (int)(sizeof(s.noise_floor) / 4))
    s.noise_floor[lane] = reinterpret_cast<const int32_t *>(p.frame[ch].int_noise_floor)[lane];
  const xaac_sbr_frame *f = reinterpret_cast<const xaac_sbr_frame *>(s.f_head); /* head members only */
  {
    int32_t *m = reinterpret_cast<int32_t *>(&s.st);
    if (lane < 2) m[lane] = gstw[kHeadOff / 4 + lane];
    copy_words(m + 2, gstw + kTailOff / 4, kTailWords, lane);
  }
  if (!HQ)
    for (int i = lane; i < 568; i += 64) s.rand_hi[i] = (int16_t)(xaac_sbr_rand_ph[i] >> 16);
  for (int i = lane; i < 256; i += 64) xs_lds_inv_table[i] = xaac_sbr_inv_table[i];
  for (int i = lane; i < 257; i += 64) xs_lds_sqrt_table[i] = xaac_sbr_sqrt_table[i];
  for (int i = lane; i < 2 * ROW; i += 64) {
    s.x[i] = 0;
    s.x[XW + (i & 127)] = 0;
  }
  copy_words(s.x + 2 * ROW, gstw + offsetof(xaac_sbr_state, overlap) / 4, 6 * ROW, lane);  /* sbr_dec.c:753 */
  for (int i0 = lane; i0 < 32 * 32 * (HQ ? 2 : 1); i0 += 64 * 8) { /* the analysed slots: bands 0..31 (re, im) */
    int32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int i = i0 + 64 * j;
      const int row = HQ ? (i >> 6) : (i >> 5), col = HQ ? ((i & 31) + ((i & 32) ? 64 : 0)) : (i & 31);
      t[j] = gx[(8 + row) * ROW + col];
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int i = i0 + 64 * j;
      const int row = HQ ? (i >> 6) : (i >> 5), col = HQ ? ((i & 31) + ((i & 32) ? 64 : 0)) : (i & 31);
      s.x[(8 + row) * ROW + col] = t[j];
    }
  }
  __syncthreads();
#ifdef XS_PROFILE
  if (lane == 0) {
    for (int i = 0; i < 32; i++) xs_prof_acc[i] = 0;
    xs_prof_last = clock64();
  }
#endif

  const XsCx cx = {lane, 64};
  const XsQmfT<HQ> x = {s.x};
  if (lane == 0) s.st.lb_scale = 0;
  const int refused = xs_side_info_bad(cx, &s.h, f, &s.st); /* counts / band numbers past the structs' capacity */
  if (f->apply_processing && !refused) xs_rescale_x_overlap(cx, &s.h, f, &s.st, x);
  /* what ixheaacd_cplx_anal_qmffilt leaves in the scale struct (generic:630-631) */
  __syncthreads();
  if (lane == 0) {
    s.st.st_lb_scale = 0;
    s.st.lb_scale = HQ ? -8 : -10;
  }
  __syncthreads();
  int save_lb_scale = 0;
#ifdef XS_SKIP_CORE
  const int rc = 0;
#else
  const int rc = refused ? -1
                         : xs_sbr_core(cx, &s.h, f, p.frame[ch].int_env_sf_arr, reinterpret_cast<const int16_t *>(s.noise_floor),
                                       &s.st, x, &s.w, s.rand_hi, &save_lb_scale);
#endif
  __syncthreads();
#ifdef XS_PROFILE
  XS_T(15);
  if (lane < 32 && p.status) atomicAdd(reinterpret_cast<unsigned long long *>(p.status) + lane, (unsigned long long)xs_prof_acc[lane]);
#endif

  /* ---- copy-out ---- */
  if (lane == 0) {
    int16_t *par = p.syn_par + 8 * (size_t)ch;
    par[0] = s.st.lb_scale;
    par[1] = s.st.ov_lb_scale;
    par[2] = s.st.hb_scale;
    par[3] = s.st.st_syn_scale;
    par[4] = s.st.syn_lsb;
    par[5] = s.st.syn_usb;
    par[6] = 0; /* channel active (the synthesis kernel skips channels flagged here) */
    par[7] = 0;
    s.st.ov_lb_scale = (int16_t)save_lb_scale;  /* sbr_dec.c:1304 */
#ifndef XS_PROFILE
    if (p.status) p.status[ch] = rc;
#endif
  }
  __syncthreads();
  copy_words(gx + 2 * ROW, s.x + 2 * ROW, 38 * ROW, lane); /* slots 0..31 for synthesis (+ 32..37 for PS) */
  {
    int32_t *gw = reinterpret_cast<int32_t *>(gst);
    /* sbr_dec.c:1283-1291 copies 6 * 64 words in either mode: in HQ the first three of the six slots */
    copy_words(gw + offsetof(xaac_sbr_state, overlap) / 4, s.x + (2 + 32) * ROW, 6 * 64, lane);
    const int32_t *m = reinterpret_cast<const int32_t *>(&s.st);
    if (lane < 2) gw[kHeadOff / 4 + lane] = m[lane];
    copy_words(gw + kTailOff / 4, m + 2, kTailWords, lane);
  }
}

extern "C" hipError_t xaac_launch_sbr_core_lp(const XaacSbrCoreParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_sbr_core_kernel<0>, dim3(p->n_ch), dim3(64), 0, stream, *p);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_sbr_core_hq(const XaacSbrCoreParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_sbr_core_kernel<1>, dim3(p->n_ch), dim3(64), 0, stream, *p);
  return hipGetLastError();
}
