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
#include <stdlib.h>

#ifdef XS_PROFILE
/* phase timers (tools/prof_sbr_core.py): cycles of each wave's lane 0 between XS_T hooks, summed over channels */
__shared__ long long xs_prof_last[4];
__shared__ unsigned xs_prof_acc[4][32]; /* per stream: fits 32 bits (and two workgroups still fit a CU's LDS) */
#define XS_T(i)                                                  \
  do {                                                           \
    if ((threadIdx.x & 63) == 0) {                               \
      long long t_ = clock64();                                  \
      xs_prof_acc[threadIdx.x >> 6][i] += (unsigned)(t_ - xs_prof_last[threadIdx.x >> 6]); \
      xs_prof_last[threadIdx.x >> 6] = t_;                       \
    }                                                            \
  } while (0)
#endif
/* LDS copies of the two tables behind every pseudo-float divide / square root (sbr_core.h: XS_TAB_*): a lookup in
   global memory on those serial paths costs its latency each time */
__shared__ int16_t xs_lds_inv_table[256];
__shared__ int16_t xs_lds_sqrt_table[258];
/* the slot loops' random phases: the 32-bit table for the HQ one (sbr_core.h: XS_TAB_RAND), its high halves for the
   low-power one -- a kernel only allocates the one it references.  All three tables are shared by a workgroup's waves */
__shared__ int32_t xs_lds_rand_ph[568];
__shared__ int16_t xs_lds_rand_hi[568];
/* ... and the four small ones (186 bytes): limiter gains [8], smoothing filter [4], 1 / n [49] | chirp targets [16] */
__shared__ int16_t xs_lds_small16[8 + 4 + 50];
__shared__ int32_t xs_lds_new_bw[16];
#define XS_TAB_LIMG(i) xs_lds_small16[i]
#define XS_TAB_SMOOTH(i) xs_lds_small16[8 + (i)]
#define XS_TAB_INVINT(i) xs_lds_small16[12 + (i)]
#define XS_TAB_NEWBW(i) xs_lds_new_bw[i]
#define XS_TAB_RAND(i) xs_lds_rand_ph[i]
#define XS_SYNC_WAVE_LDS 1 /* XsCx::sync(): wave-level, LDS only (see sbr_core.h) */
#define XS_TAB_INV(i) xs_lds_inv_table[i]
#define XS_TAB_SQRT(i) xs_lds_sqrt_table[i]
#include "sbr_core.h"
#include "sbr_core_kernel.h"

#ifndef XS_STAGGER_G
#define XS_STAGGER_G 64     /* wave groups of the persistent launch's staggered start (1: none) */
#endif
#ifndef XS_STAGGER_SLEEP
#define XS_STAGGER_SLEEP 16 /* x 64 cycles between neighbouring groups */
#endif

namespace {

__device__ __forceinline__ void xs_wave_sync() { /* = XsCx::sync() */
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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

/* NB: bands a matrix row holds in LDS ("narrow rows").  The reference's rows have 64 bands; an SBR range that ends at
   band 48 or below (sub_band_end, patches, tables, bank limits, nothing left above in the overlap slots) never touches
   the rest, and 40 x 2 x 48 words instead of 40 x 2 x 64 let eight waves share a CU's LDS instead of six -- the kernel
   is latency bound, its throughput follows the number of resident waves (measured with padded LDS: 4, 5, 6 waves per CU gave 528, 483, 348 us, docs/NOTEBOOK.md 5k).  A stream
   that does not qualify is appended to a list and runs through the 64-band instantiation in a second, list-driven
   launch: same code, same results. */
template <int HQ, int NB>
struct XsLds {
  typedef XsQmfT<HQ, NB> Q;
  int32_t x[XAAC_SBR_X_ROWS * Q::ROW + (HQ ? 0 : 128)]; /* LP: + rows the reference's edge writes may run into past slot 37 */
  XsLdsState st;
  xaac_sbr_header h;
  int32_t f_head[kFrameHeadBytes / 4];
  int32_t noise_floor[sizeof(((xaac_sbr_frame *)0)->int_noise_floor) / 4];
  XsWork w;
#ifdef XS_LDS_PAD
  char occupancy_probe[XS_LDS_PAD]; /* developer experiment: fewer resident waves */
#endif
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

/* one channel-frame; returns false when the stream has to go through the 64-band rows (nothing written then) */
template <int HQ, int NB>
__device__ __forceinline__ bool core_one(const XaacSbrCoreParams &p, const int ch, XsLds<HQ, NB> &s, const int lane) {
  typedef XsQmfT<HQ, NB> Q;
  constexpr int ROW = Q::ROW, ROWG = HQ ? 128 : 64, XWG = (HQ ? 2 : 1) * XAAC_SBR_X_WORDS;
  xaac_sbr_state *gst = p.state + ch;
  int32_t *gx = p.x + (size_t)ch * XWG;
  const int32_t *gstw = reinterpret_cast<const int32_t *>(gst);

#ifdef XS_PROFILE
  if (lane == 0) {
    for (int i = 0; i < 32; i++) xs_prof_acc[threadIdx.x >> 6][i] = 0;
    xs_prof_last[threadIdx.x >> 6] = clock64();
  }
#endif
  /* ---- copy-in: every global load of the channel-frame is issued before the first LDS store, so the wave pays one
     memory latency here, not one per piece (the kernel is latency bound: a wave's lifetime is its cost) ---- */
  constexpr int NH = (sizeof(xaac_sbr_header) / 4 + 63) / 64, NF = (kFrameHeadBytes / 4 + 63) / 64;
  constexpr int NT = (kTailWords + 63) / 64, NOV = 6 * ROWG / 64, NAS = 32 * 32 * (HQ ? 2 : 1) / 64;
  constexpr int NNF = sizeof(s.noise_floor) / 4;
  static_assert(NOV * 64 == 6 * ROWG && NF == 1, "whole rows of lanes");
  constexpr int NSF = (sizeof(((xaac_sbr_frame *)0)->int_env_sf_arr) / 4 + 63) / 64; /* the envelopes' scale factors: 224 words */
  int32_t r_h[NH], r_f, r_nf = 0, r_hd = 0, r_t[NT], r_ov[NOV], r_an[NAS], r_sf[NSF];
  {
    const int32_t *gh = reinterpret_cast<const int32_t *>(p.header + ch), *gf = reinterpret_cast<const int32_t *>(p.frame + ch);
#pragma unroll
    for (int j = 0; j < NH; j++) r_h[j] = lane + 64 * j < (int)(sizeof(xaac_sbr_header) / 4) ? gh[lane + 64 * j] : 0;
    r_f = lane < kFrameHeadBytes / 4 ? gf[lane] : 0;
    if (lane < NNF) r_nf = reinterpret_cast<const int32_t *>(p.frame[ch].int_noise_floor)[lane];
    /* (kept in registers until the envelope adjuster asks for them: read there they were a memory round trip in the middle of
       the frame, and LDS has no 896 bytes to spare) */
#pragma unroll
    for (int j = 0; j < NSF; j++)
      r_sf[j] = lane + 64 * j < (int)(sizeof(((xaac_sbr_frame *)0)->int_env_sf_arr) / 4) ? reinterpret_cast<const int32_t *>(p.frame[ch].int_env_sf_arr)[lane + 64 * j] : 0;
    if (lane < 2) r_hd = gstw[kHeadOff / 4 + lane];
#pragma unroll
    for (int j = 0; j < NT; j++) r_t[j] = lane + 64 * j < kTailWords ? gstw[kTailOff / 4 + lane + 64 * j] : 0;
    const int32_t *gov = gstw + offsetof(xaac_sbr_state, overlap) / 4; /* sbr_dec.c:753 */
#pragma unroll
    for (int j = 0; j < NOV; j++) r_ov[j] = gov[lane + 64 * j];
#pragma unroll
    for (int j = 0; j < NAS; j++) { /* the analysed slots: bands 0..31 (re, im) */
      const int i = lane + 64 * j;
      const int row = HQ ? (i >> 6) : (i >> 5), col = HQ ? ((i & 31) + ((i & 32) ? 64 : 0)) : (i & 31);
      r_an[j] = gx[(8 + row) * ROWG + col];
    }
  }
  const xaac_sbr_frame *f = reinterpret_cast<const xaac_sbr_frame *>(s.f_head); /* head members only */
  int32_t above = 0; /* OR of the overlap words in bands the narrow rows do not hold */
  {
#pragma unroll
    for (int j = 0; j < NH; j++)
      if (lane + 64 * j < (int)(sizeof(xaac_sbr_header) / 4)) reinterpret_cast<int32_t *>(&s.h)[lane + 64 * j] = r_h[j];
    if (lane < kFrameHeadBytes / 4) s.f_head[lane] = r_f;
    if (lane < NNF) s.noise_floor[lane] = r_nf;
    int32_t *m = reinterpret_cast<int32_t *>(&s.st);
    if (lane < 2) m[lane] = r_hd;
#pragma unroll
    for (int j = 0; j < NT; j++)
      if (lane + 64 * j < kTailWords) m[2 + lane + 64 * j] = r_t[j];
    for (int i = lane; i < 2 * ROW; i += 64) s.x[i] = 0;
    if (!HQ)
      for (int i = lane; i < 128; i += 64) s.x[XAAC_SBR_X_ROWS * ROW + i] = 0;
  }
  /* a run of 64 words of a global row is one part (real | imaginary) of one slot with band = lane (HQ), or one slot
     (LP): the LDS place follows from j and the lane with one predicate, band < NB.  The matrix words stay in their registers
     until the block floating point of the low bands has been through them (below). */
  const bool held = NB == 64 || lane < NB;
#pragma unroll
  for (int j = 0; j < NOV; j++) above |= held ? 0 : r_ov[j];
  auto store_matrix = [&]() {
    if (held) { /* one predicated region for the twelve stores */
#pragma unroll
      for (int j = 0; j < NOV; j++) {
        const int row = HQ ? j >> 1 : j, part = HQ ? j & 1 : 0;
        s.x[(2 + row) * ROW + part * NB + lane] = r_ov[j];
      }
    }
#pragma unroll
    for (int j = 0; j < NAS; j++) {
      const int i = lane + 64 * j;
      const int row = HQ ? (i >> 6) : (i >> 5), col = HQ ? ((i & 31) + ((i & 32) ? Q::IM : 0)) : (i & 31);
      s.x[(8 + row) * ROW + col] = r_an[j];
    }
    /* what the analysis bank does not write is 0: bands 32 and up of the analysed slots (sbr_dec.c:1121) */
    for (int i = lane; i < 32 * (HQ ? 2 : 1) * (NB - 32); i += 64) {
      const int row = i / ((HQ ? 2 : 1) * (NB - 32)), c = i % ((HQ ? 2 : 1) * (NB - 32));
      s.x[(8 + row) * ROW + (c < NB - 32 ? 32 + c : Q::IM + 32 + (c - (NB - 32)))] = 0;
    }
  };
  xs_wave_sync();
#ifdef XS_PROFILE
  XS_T(24);
#endif

  const XsCx cx = {lane, 64};
  const Q x = {s.x};
  if (lane == 0) s.st.lb_scale = 0;
  const int refused = xs_side_info_bad(cx, &s.h, f, &s.st); /* counts / band numbers past the structs' capacity */
  if (NB < 64) {
    /* every band number this frame can turn into a column: the SBR range and its tables, the patches' targets, the
       bank limits xs_rescale_x_overlap walks between (all within 0..64 once xs_side_info_bad has passed) */
    int32_t top = 0;
    if (!refused) {
      const int32_t lim[4] = {s.st.syn_usb, s.st.syn_lsb, s.st.codec_usb, s.st.prev_max_qmf_subband_aac};
      for (int i = 0; i < 4; i++) top = lim[i] > top ? lim[i] : top;
      if (f->apply_processing) { /* (a frame without SBR processing reads none of the header's or the frame's band numbers) */
        top = s.h.sub_band_end > top ? s.h.sub_band_end : top;
        top = f->max_qmf_subband_aac > top ? f->max_qmf_subband_aac : top;
        if (lane <= s.h.num_sf_bands[1] && s.h.freq_band_tbl_hi[lane] > top) top = s.h.freq_band_tbl_hi[lane];
        if (lane <= s.h.num_sf_bands[0] && s.h.freq_band_tbl_lo[lane] > top) top = s.h.freq_band_tbl_lo[lane];
        if (lane < s.h.num_patches) {
          const xaac_sbr_patch *pp = &s.h.patch[lane];
          const int a = pp->src_end_band + pp->dst_end_band, b = pp->dst_start_band + pp->num_bands_in_patch;
          top = a > top ? a : top;
          top = b > top ? b : top;
        }
      }
    }
    if (cx.wave_max(top) > NB || cx.wave_or(above) != 0) return false;
  }
  /* The block floating point of the low bands (sbr_dec.c:1050-1120: headroom of the overlap slots and of the analysed slots,
     two shifts) is taken from, and applied to, the words while they are in registers -- as two scans and two read-modify-write
     walks over the matrix in LDS it was a tenth of the kernel.  A frame whose overlap slots xs_rescale_x_overlap has to walk
     first (the cross-over band moved: rare) goes through LDS as before, and so does a refused one. */
  const int apply = __builtin_amdgcn_readfirstlane((int)f->apply_processing);
  const int old_lsb = __builtin_amdgcn_readfirstlane((int)s.st.prev_max_qmf_subband_aac);
  const int new_lsb = __builtin_amdgcn_readfirstlane((int)f->max_qmf_subband_aac);
  const bool in_regs = !refused && !(apply && new_lsb != old_lsb && old_lsb > 0);
  if (!in_regs) store_matrix();
  if (apply && !refused) xs_rescale_x_overlap(cx, &s.h, f, &s.st, x); /* in_regs: only its two state words */
  /* what ixheaacd_cplx_anal_qmffilt leaves in the scale struct (generic:630-631) */
  xs_wave_sync();
  if (lane == 0) {
    s.st.st_lb_scale = 0;
    s.st.lb_scale = HQ ? -8 : -10;
  }
  xs_wave_sync();
  int save_lb_scale = 0;
  XsPendingAdjust pend = {0, 0, 0, 0, 0}; /* the envelope adjuster's last two shifts, applied on the way out (copy-out) */
  XS_T(26);
#ifdef XS_SKIP_CORE
  const int rc = 0;
  if (in_regs) store_matrix();
#else
  int rc = -1;
  if (in_regs) {
    const int usb = __builtin_amdgcn_readfirstlane((int)s.st.codec_usb);
    int32_t m_ov = 1, m_an = 1;
    const bool low_ov = lane < usb, low_an = (lane & 31) < usb;
#pragma unroll
    for (int j = 0; j < NOV; j++) m_ov |= low_ov ? fx_abs_nrm(r_ov[j]) : 0;
#pragma unroll
    for (int j = 0; j < NAS; j++) m_an |= low_an ? fx_abs_nrm(r_an[j]) : 0;
    const int reserve = xs_pnorm32(cx.wave_or(m_an)), reserve_ov1 = xs_pnorm32(cx.wave_or(m_ov));
    const XsBfp b = xs_bfp_shifts<HQ>(cx, &s.st, usb, reserve, reserve_ov1);
    if (b.sh_ov != 0) {
#pragma unroll
      for (int j = 0; j < NOV; j++) r_ov[j] = low_ov ? xs_adjust_word(r_ov[j], b.sh_ov) : r_ov[j];
    }
    if (b.sh_main != 0) {
#pragma unroll
      for (int j = 0; j < NAS; j++) r_an[j] = low_an ? xs_adjust_word(r_an[j], b.sh_main) : r_an[j];
    }
    store_matrix();
    save_lb_scale = b.save_lb_scale;
    xs_wave_sync();
    XS_T(1);
    rc = xs_sbr_core_tail(cx, &s.h, f, p.frame[ch].int_env_sf_arr, reinterpret_cast<const int16_t *>(s.noise_floor), &s.st, x,
                          &s.w, HQ ? nullptr : xs_lds_rand_hi, b, &pend, r_sf);
  } else if (!refused) {
    rc = xs_sbr_core(cx, &s.h, f, p.frame[ch].int_env_sf_arr, reinterpret_cast<const int16_t *>(s.noise_floor), &s.st, x, &s.w,
                     HQ ? nullptr : xs_lds_rand_hi, &save_lb_scale);
  }
#endif
  xs_wave_sync();
#ifdef XS_PROFILE
  XS_T(15);
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
  xs_wave_sync();
  /* slots 0..31 for synthesis (+ 32..37 for PS); bands the narrow rows do not hold are 0 */
  {
    constexpr int NW = 38 * ROWG / 64; /* runs of 64 words: (slot, part) with band = lane */
    const bool held = NB == 64 || lane < NB;
    const int lane_h = held ? lane : 0; /* every lane reads (a select, not a predicated region per word) */
    /* env_calc.c:975-1003 on the way: the adjusted bands' slots below 32 take the shift xs_calc_sbrenvelope left pending (a
       left count and a right count of which one is zero, per slot range) */
    const bool adj_band = lane >= pend.b0 && lane < pend.b1;
    const auto counts = [](int sh, int &l, int &r) {
      sh = sh > 31 ? 31 : (sh < -31 ? -31 : sh);
      l = sh > 0 ? sh : 0;
      r = sh < 0 ? -sh : 0;
    };
    int ov_l, ov_r, mn_l, mn_r;
    counts(pend.sh_ov, ov_l, ov_r);
    counts(pend.sh_main, mn_l, mn_r);
#pragma unroll
    for (int j0 = 0; j0 < NW; j0 += 8) {
      int32_t t[8];
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (j0 + j < NW) {
          const int row = HQ ? (j0 + j) >> 1 : j0 + j, part = HQ ? (j0 + j) & 1 : 0;
          int32_t v = s.x[(2 + row) * ROW + part * NB + lane_h];
          if (row < 32) {
            const int sl = row < pend.first_start ? ov_l : mn_l, sr = row < pend.first_start ? ov_r : mn_r; /* (uniform) */
            const int32_t w = (int32_t)((uint32_t)v << sl) >> sr;
            v = adj_band ? w : v;
          }
          t[j] = held ? v : 0;
        }
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (j0 + j < NW) gx[2 * ROWG + 64 * (j0 + j) + lane] = t[j];
    }
  }
  {
    int32_t *gw = reinterpret_cast<int32_t *>(gst);
    /* sbr_dec.c:1283-1291 copies 6 * 64 words in either mode: in HQ the first three of the six slots */
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const int row = HQ ? j >> 1 : j, part = HQ ? j & 1 : 0;
      const bool held = NB == 64 || lane < NB;
      int32_t v = s.x[(2 + 32 + row) * ROW + part * NB + (held ? lane : 0)];
      /* (a first border behind slot 32 -- no parser's grid has one -- leaves the pending overlap-side shift to these rows too:
         env_calc.c:975 adjusts slots 0 .. first border) */
      if (32 + row < pend.first_start && lane >= pend.b0 && lane < pend.b1) {
        const int sh = pend.sh_ov > 31 ? 31 : (pend.sh_ov < -31 ? -31 : pend.sh_ov);
        v = sh > 0 ? (int32_t)((uint32_t)v << sh) : (v >> -sh);
      }
      gw[offsetof(xaac_sbr_state, overlap) / 4 + 64 * j + lane] = held ? v : 0;
    }
    const int32_t *m = reinterpret_cast<const int32_t *>(&s.st);
    if (lane < 2) gw[kHeadOff / 4 + lane] = m[lane];
    copy_words(gw + kTailOff / 4, m + 2, kTailWords, lane);
  }
#ifdef XS_PROFILE
  XS_T(25);
  xs_wave_sync();
  if (lane < 32 && p.status) atomicAdd(reinterpret_cast<unsigned long long *>(p.status) + lane, (unsigned long long)xs_prof_acc[threadIdx.x >> 6][lane]);
#endif
  return true;
}

/* the workgroup's tables, staged by all its threads (loads first, then the LDS stores) */
template <int HQ, int THREADS>
__device__ __forceinline__ void stage_tables(int tid) {
  constexpr int NR = (568 + THREADS - 1) / THREADS, NI = (256 + THREADS - 1) / THREADS, NS = (257 + THREADS - 1) / THREADS;
  int32_t tr[NR];
  int16_t ti[NI], ts[NS];
  /* every load unconditional (an index past a table's end reads its last entry and the value is dropped): as `in range ? load : 0`
     each became a branch around a load with its own wait -- ten memory round trips one after the other at the start of every
     workgroup, which the one-shot low-power launch pays per pair of channel-frames */
  const auto upto = [](int i, int n) { return i < n ? i : n - 1; };
#pragma unroll
  for (int j = 0; j < NR; j++) tr[j] = xaac_sbr_rand_ph[upto(tid + THREADS * j, 568)];
#pragma unroll
  for (int j = 0; j < NI; j++) ti[j] = xaac_sbr_inv_table[upto(tid + THREADS * j, 256)];
#pragma unroll
  for (int j = 0; j < NS; j++) ts[j] = xaac_sbr_sqrt_table[upto(tid + THREADS * j, 257)];
  const int16_t s_lim = xaac_sbr_lim_gains_m[upto(tid, 8)], s_smooth = xaac_sbr_smooth_filter[upto(tid - 8 < 0 ? 0 : tid - 8, 4)];
  const int16_t s_inv = xaac_sbr_inv_int_table[upto(tid - 12 < 0 ? 0 : tid - 12, 49)];
  const int32_t s_bw = xaac_sbr_new_bw_table[upto(tid, 16)];
  if (tid < 8) xs_lds_small16[tid] = s_lim;
  if (tid >= 8 && tid < 12) xs_lds_small16[tid] = s_smooth;
  if (tid >= 12 && tid < 12 + 49) xs_lds_small16[tid] = s_inv;
  if (tid < 16) xs_lds_new_bw[tid] = s_bw;
#pragma unroll
  for (int j = 0; j < NR; j++)
    if (tid + THREADS * j < 568) {
      if (HQ)
        xs_lds_rand_ph[tid + THREADS * j] = tr[j];
      else
        xs_lds_rand_hi[tid + THREADS * j] = (int16_t)(tr[j] >> 16);
    }
#pragma unroll
  for (int j = 0; j < NI; j++)
    if (tid + THREADS * j < 256) xs_lds_inv_table[tid + THREADS * j] = ti[j];
#pragma unroll
  for (int j = 0; j < NS; j++)
    if (tid + THREADS * j < 257) xs_lds_sqrt_table[tid + THREADS * j] = ts[j];
}

}  // namespace

/* HQ = 0: low-power mode, rows of 64 reals; HQ = 1: rows of NB real | NB imaginary (HE-AAC mono / v2).
   WAVES waves per workgroup share the tables; every wave takes channel-frames off a work counter until none is left
   (p.work_counter, zeroed by the launch; without one: workgroup i's wave w takes channel WAVES i + w).  With NB < 64 a
   stream that needs the full rows is appended to p.defer_list. */
/* The low-power instantiation is asked to fit three waves per SIMD (168 VGPRs; it takes 189 with the two-envelope passes
   otherwise, which leaves four of the five workgroups a CU's LDS holds: 326 us per C3 step; capped, without spills, 299) */
#ifndef XS_LP_MIN_WAVES
#define XS_LP_MIN_WAVES 3
#endif
template <int HQ, int NB, int WAVES>
__global__ __launch_bounds__(64 * WAVES, HQ ? 1 : XS_LP_MIN_WAVES) void xaac_sbr_core_kernel(XaacSbrCoreParams p) {
  __shared__ XsLds<HQ, NB> s[WAVES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  stage_tables<HQ, 64 * WAVES>(threadIdx.x);
  __syncthreads();
  /* Staggered start (persistent launch only).  The resident waves all start together and a channel-frame takes every wave
     about the same time, so their copy phases -- 13 KB in, 22 KB out per channel-frame -- come as chip-wide bursts: for those
     microseconds the memory system runs at its limit and nothing computes, then everything computes and the memory system
     idles (phase timers: copy-in + copy-out = 24 % of a wave's life for 2 % of its instructions).  Wave group g of
     XS_STAGGER_G starts g x XS_STAGGER_SLEEP x 64 cycles late (64 groups x 0.43 us: the last one 27 us behind the first, two
     thirds of a channel-frame's time), which spreads the phases for the launch's four rounds; the work counter evens out
     what the late starters do not get to.  Measured on one box (profiles/r05_b_stagger.txt): none 189.7 us; 2 groups x 6.8 us
     187; 4 x 3.4 us 181.6; 8 x 3.4 us 175.7; 32 x 0.85 us 175.3; 64 x 0.43 us 172.8; 8 x 6.8 us 186; 16 x 3.4 us 185.
     The one-shot low-power launch (6.4 rounds of workgroups that start as others end) only loses by it: 404 -> 415-436 us. */
  if (XS_STAGGER_G > 1 && p.work_counter && p.n_ch >= 3 * (int)gridDim.x * WAVES) { /* (a small batch has no second round to gain in) */
    const int g = ((int)blockIdx.x * WAVES + wave) % XS_STAGGER_G;
    for (int t = 0; t < g; t++) __builtin_amdgcn_s_sleep(XS_STAGGER_SLEEP);
  }
  for (bool first = true;; first = false) { /* one call site: one copy of the core in the kernel's code */
    int ch = 0;
    if (p.work_counter) {
      if (lane == 0) ch = atomicAdd(p.work_counter, 1);
      ch = __builtin_amdgcn_readfirstlane(ch);
    } else {
      ch = first ? (int)blockIdx.x * WAVES + wave : p.n_ch;
    }
    if (ch >= p.n_ch) break;
    if (!core_one<HQ, NB>(p, ch, s[wave], lane) && lane == 0) {
      if (p.narrow_only) { /* the hint was wrong for this stream: refused, nothing of it written */
        if (p.status) p.status[ch] = XAAC_FATAL_BAD_ARG;
      } else {
        p.defer_list[atomicAdd(p.defer_count, 1)] = ch;
      }
    }
    xs_wave_sync();
  }
}

/* the streams of p.defer_list through the 64-band rows: a small grid walks the list (usually empty) */
template <int HQ>
__global__ __launch_bounds__(64) void xaac_sbr_core_list_kernel(XaacSbrCoreParams p) {
  __shared__ XsLds<HQ, 64> s;
  const int lane = threadIdx.x;
  const int n = *p.defer_count;
  if ((int)blockIdx.x >= n) return;
  stage_tables<HQ, 64>(lane);
  for (int j = blockIdx.x; j < n; j += gridDim.x) {
    xs_wave_sync();
    core_one<HQ, 64>(p, p.defer_list[j], s, lane);
  }
}

extern "C" hipError_t xaac_launch_sbr_core_lp(const XaacSbrCoreParams *p, hipStream_t stream) {
  XaacSbrCoreParams q = *p;
  q.work_counter = nullptr;
#ifndef XS_LP_WAVES
#define XS_LP_WAVES 2 /* measured: 1: 457 us, 2: 410 (30 KB of LDS per workgroup: ten waves per CU instead of nine), 3: 454 */
#endif
  constexpr int W = XS_LP_WAVES; /* waves (channel-frames) per workgroup: they share one staging of the tables */
#ifndef XS_LP_PERSISTENT
#define XS_LP_PERSISTENT 1
#endif
  if (XS_LP_PERSISTENT && p->work_counter && p->counters_zeroed) {
    /* persistent, like the HQ launch: as many workgroups as the chip holds (five per CU: 31.6 KB of LDS each), channel-frames
       off the work counter, the tables staged once per workgroup instead of once per pair of channel-frames, staggered start */
#ifndef XS_LP_WG_PER_CU
#define XS_LP_WG_PER_CU 5
#endif
    const int resident = XS_LP_WG_PER_CU * (p->num_cu > 0 ? p->num_cu : 256), need = (p->n_ch + W - 1) / W;
    hipLaunchKernelGGL((xaac_sbr_core_kernel<0, 64, W>), dim3(need < resident ? need : resident), dim3(64 * W), 0, stream, *p);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((xaac_sbr_core_kernel<0, 64, W>), dim3((p->n_ch + W - 1) / W), dim3(64 * W), 0, stream, q);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_sbr_core_hq(const XaacSbrCoreParams *p, hipStream_t stream) {
  if (!p->defer_list || !p->defer_count || !p->work_counter) {
    XaacSbrCoreParams q = *p;
    q.work_counter = nullptr;
    hipLaunchKernelGGL((xaac_sbr_core_kernel<1, 64, 1>), dim3(p->n_ch), dim3(64), 0, stream, q);
    return hipGetLastError();
  }
  /* defer_count and work_counter are neighbours */
  if (!p->counters_zeroed) {
    hipError_t e = hipMemsetAsync(p->defer_count, 0, 2 * sizeof(int32_t), stream);
    if (e != hipSuccess) return e;
  }
  constexpr int W = XAAC_SBR_CORE_HQ_WAVES;
  static int per_cu = 0; /* developer override: XAAC_CORE_WG_PER_CU (two fill a CU's LDS) */
  if (!per_cu) {
    const char *e = getenv("XAAC_CORE_WG_PER_CU");
    per_cu = e && atoi(e) > 0 ? atoi(e) : 2;
  }
  const int resident = per_cu * (p->num_cu > 0 ? p->num_cu : 256), need = (p->n_ch + W - 1) / W;
  hipLaunchKernelGGL((xaac_sbr_core_kernel<1, XAAC_SBR_NARROW_BANDS, W>), dim3(need < resident ? need : resident), dim3(64 * W), 0,
                     stream, *p);
  if (p->narrow_only) return hipGetLastError(); /* the caller knows the list stays empty (xaac_sbr_hq_batch.max_band_hint) */
  const int grid = p->n_ch < 64 ? p->n_ch : 64;
  hipLaunchKernelGGL((xaac_sbr_core_list_kernel<1>), dim3(grid), dim3(64), 0, stream, *p);
  return hipGetLastError();
}

/* xaac_warm_up (xaac_abi.cpp): asking for a kernel's attributes puts this translation unit's code object on the device */
extern "C" hipError_t xaac_warm_sbr_core(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&xaac_sbr_core_kernel<1, XAAC_SBR_NARROW_BANDS, XAAC_SBR_CORE_HQ_WAVES>));
}
