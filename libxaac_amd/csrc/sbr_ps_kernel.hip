/*
 * sbr_ps_kernel.hip -- the parametric-stereo tool of HE-AACv2 on gfx950: what the reference runs inside the
 * left channel's synthesis loop (decoder/ixheaacd_qmf_dec.c:1015-1031 with ixheaacd_thumb_ps_dec.c:69 and
 * ixheaacd_ps_dec.c), pulled out in front of the two synthesis banks.
 *
 * Mapping: ONE WAVE = ONE STREAM-FRAME, persistent workgroups of one wave (the PS constants are staged in LDS once per
 * workgroup, then the wave walks over its streams).  The frame is not walked slot by slot: sbr_ps_frame.h runs it in
 * phases -- everything that depends on its slot only is spread over the lanes (QMF bands with coalesced row
 * accesses, (slot, sub-band) pairs, (slot, bin) pairs), and the three true recursions over the slots (transient
 * detector, all-pass chains, envelope counter) run as short lane-parallel loops over pre-computed inputs.  The first
 * version of this kernel walked the slots with every phase inside the loop: 25.7 k VALU wave-instructions per
 * stream-frame, most of them with a few active lanes (395 us per 8192 streams, profiles/r01_m_c4_kernel_stats.txt).
 * The PS state (5.2 KB), the frame's side info and the phase buffers (17 KB) live in LDS; the QMF matrix stays where
 * the core kernel put it: 38 rows are read (twice: powers, then rotation), 32 + 32 rows written.  The kernel also
 * does what ixheaacd_cplx_synt_qmffilt does around the tool -- bring the matrix to the PS scale (adjust_scale,
 * qmf_dec.c:937), the hybrid look-ahead's scale (thumb:77), the common shift in front of the left bank
 * (generic:1610) -- so that both synthesis launches find their input ready.
 */
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

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
/* XsCx::sync(): a stream-frame is one wave and what its lanes share is in LDS (matrix words are only ever read back by
   the lane that wrote them), so producer / consumer order inside the wave is all it takes -- see sbr_core.h */
#define XS_SYNC_WAVE_LDS 1
#include "sbr_ps_frame.h"
#include "sbr_ps_kernel.h"

namespace {

struct XpLdsState {
  XAAC_PS_STATE_HEAD_FIELDS
};
constexpr int kHeadWords = offsetof(xaac_ps_state, syn_ring_r) / 4;
static_assert(offsetof(xaac_ps_state, syn_ring_r) % 4 == 0 && sizeof(XpLdsState) == kHeadWords * 4, "mirror layout");
static_assert(sizeof(xaac_ps_frame) % 4 == 0, "word copies");

/* the PS constants, all of them (a table lookup in global memory costs a serial phase its latency; the quarter-wave sine table
   of the envelope borders' coefficient set-up stayed in global memory until the LDS budget was counted: 2 x (4 x 17.9 KB +
   2.4 KB) = 148 KB per CU) */
constexpr int kTabBytes = (int)sizeof(XpTables);
#ifndef XP_WAVES
#define XP_WAVES 4
#endif
#ifndef XP_WAVES_PER_EU
#define XP_WAVES_PER_EU 2
#endif
constexpr int kPsWaves = XP_WAVES; /* waves (stream-frames in flight) per workgroup: they share the tables' LDS copy */
struct XpLds {
  XpLdsState ps;
  xaac_ps_frame pf;
  XpFrameWork w;
};

__device__ __forceinline__ void xp_wave_sync() { /* = XsCx::sync() */
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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

/* xp_init_ps_scale (sbr_ps.h; ps_dec.c:134-210, thumb_ps_dec.c:101) on the state's words while they are in registers between
   global memory and LDS: word w of the head is in rs[w / 64] of lane w % 64.  Headroom of the delay lines (16-bit pairs) and the
   hybrid filter history (32-bit), the scale that follows from it and the core's three scales, and the shift applied to every
   member xp_scale_states names -- as passes over the LDS copy (2-byte accesses) this was an eighth of the kernel.  Which
   words: ser[j][i][6..45] for j < sample_ser[i], ap[m][6..45], ld, sd[0..57], sub, sub_ser (pairs); peak_decay_diff /
   energy_prev / peak_decay_diff_prev (32-bit, twice the shift, not in the headroom); hyb_buf (32-bit). */
template <int NS>
__device__ __forceinline__ int xp_init_ps_scale_regs(const XsCx &cx, int32_t (&rs)[NS], int lb_scale, int ov_lb_scale, int hb_scale) {
  constexpr int W_AP = offsetof(XpLdsState, ap) / 4, W_LD = offsetof(XpLdsState, ld) / 4, W_SD = offsetof(XpLdsState, sd) / 4;
  constexpr int W_SUB = offsetof(XpLdsState, sub) / 4, W_IDX = offsetof(XpLdsState, idx_ser) / 4;
  constexpr int W_PK = offsetof(XpLdsState, peak_decay_diff) / 4, W_HYB = offsetof(XpLdsState, hyb_buf) / 4;
  constexpr int W_H = offsetof(XpLdsState, h11_h12_vec) / 4, W_DBS = offsetof(XpLdsState, delay_buffer_scale) / 4;
  static_assert(offsetof(XpLdsState, ser) == 0 && W_AP == 480 && W_LD == W_AP + 64 && W_SD == W_LD + 168 && W_SUB == W_SD + 32, "layout");
  static_assert(W_IDX == W_SUB + 32 + 240 && offsetof(XpLdsState, sample_ser) == 4 * W_IDX + 6 && W_PK == W_IDX + 4, "layout");
  static_assert(W_HYB == W_PK + 60 && W_H == W_HYB + 72 && offsetof(XpLdsState, delay_buffer_scale) % 4 == 0 && W_DBS < 64 * NS, "layout");
  const auto word_of = [&](int w) { return __builtin_amdgcn_readlane(rs[w / 64], w % 64); }; /* w: a constant */
  /* sample_ser[0] is the high half of word W_IDX + 1, [1] and [2] are word W_IDX + 2 */
  const int ss0 = word_of(W_IDX + 1) >> 16, ss1 = (int16_t)word_of(W_IDX + 2), ss2 = word_of(W_IDX + 2) >> 16;
  const auto is_pair = [&](int w) -> bool { /* a pair of 16-bit delay-line samples that takes part */
    const int k = w & 31, blk = w >> 5;
    const bool mid = k >= 3 && k <= 22; /* elements 6..45 of a row of 64 */
    if (w < W_AP) {
      const int i = blk % 3, j = blk / 3;
      return mid && j < (i == 0 ? ss0 : (i == 1 ? ss1 : ss2));
    }
    if (w < W_LD) return mid;
    if (w < W_SD + 29) return true;
    if (w < W_SUB) return false;
    return w < W_IDX;
  };
  int32_t m16 = 0, m32 = 0;
#pragma unroll
  for (int j = 0; j < NS; j++) {
    const int w = cx.lane + 64 * j;
    const int32_t v = rs[j], lo = (int16_t)v, hi = v >> 16;
    m16 |= is_pair(w) ? (fx_abs_nrm(lo) | fx_abs_nrm(hi)) : 0;
    if (64 * j + 63 >= W_HYB && 64 * j < W_H) m32 |= (w >= W_HYB && w < W_H) ? fx_abs_nrm(v) : 0;
  }
  const int reserve = xs_pnorm32(cx.wave_or((int32_t)((uint32_t)m16 << 16) | m32));
  const int dbs = (int16_t)((int16_t)word_of(W_DBS) + reserve);
  int16_t t = (int16_t)(lb_scale < ov_lb_scale ? lb_scale : ov_lb_scale);
  if (hb_scale < t) t = (int16_t)hb_scale;
  if (dbs < t) t = (int16_t)dbs;
  const int ps_scale = t - 1;
  const int scale = (int16_t)((ps_scale - dbs) + reserve);
  if (scale != 0) { /* (uniform) */
    const int l16 = scale > 0 ? (scale > 15 ? 15 : scale) : 0, r16 = scale < 0 ? (-scale > 31 ? 31 : -scale) : 0;
#pragma unroll
    for (int j = 0; j < NS; j++) {
      const int w = cx.lane + 64 * j;
      const int32_t v = rs[j];
      if (64 * j < W_IDX) { /* xp_scale16 on both halves: left saturating, right arithmetic; one of the counts is 0 */
        const int32_t lo = (int16_t)v, hi = v >> 16;
        const int16_t a = fx_sat16(xs_shl(lo, l16) >> r16), b = fx_sat16(xs_shl(hi, l16) >> r16);
        rs[j] = is_pair(w) ? (int32_t)xp_pack16(a, b) : v;
      }
      if (64 * j + 63 >= W_PK && 64 * j < W_H) { /* xp_scale32: the detector's three arrays by 2 x scale, the filter history by scale */
        const bool pk = w >= W_PK && w < W_HYB, hy = w >= W_HYB && w < W_H;
        const int sc = pk ? 2 * scale : scale;
        const int32_t r = sc > 0 ? fx_shl_sat(rs[j], sc) : fx_shr(rs[j], -sc);
        rs[j] = (pk || hy) ? r : rs[j];
      }
    }
  }
  /* ps->delay_buffer_scale = ps_scale: the low half of word W_DBS */
  if (cx.lane == W_DBS % 64) rs[W_DBS / 64] = (int32_t)((uint32_t)rs[W_DBS / 64] & 0xffff0000u) | (uint16_t)(int16_t)ps_scale;
  return ps_scale;
}

/* xp_frame_sanitize (sbr_ps.h) the same way, on the side info's words in registers: borders into 0..32, IID indices into
   +-7 (+-15 fine), ICC indices into 0..7; returns whether anything had to be changed */
template <int NF>
__device__ __forceinline__ int xp_frame_sanitize_regs(const XsCx &cx, int32_t (&rf)[NF]) {
  constexpr int E_BORDER = offsetof(xaac_ps_frame, border_position) / 2, E_IID = offsetof(xaac_ps_frame, iid_par_table) / 2;
  constexpr int E_ICC = offsetof(xaac_ps_frame, icc_par_table) / 2, E_END = sizeof(xaac_ps_frame) / 2;
  static_assert(offsetof(xaac_ps_frame, iid_quant) == 0 && E_BORDER + XAAC_PS_MAX_ENV + 2 < E_IID && E_IID % 2 == 0 && E_ICC % 2 == 0, "layout");
  const int steps = (int16_t)__builtin_amdgcn_readlane(rf[0], 0) ? 15 : 7;
  int bad = 0;
#pragma unroll
  for (int j = 0; j < NF; j++) {
    const int w = cx.lane + 64 * j;
    const int32_t v = rf[j];
    int h[2] = {(int16_t)v, v >> 16};
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int e = 2 * w + c; /* element number */
      const bool border = e >= E_BORDER && e < E_BORDER + XAAC_PS_MAX_ENV + 2, iid = e >= E_IID && e < E_ICC, icc = e >= E_ICC && e < E_END;
      const int lo = border ? 0 : (iid ? -steps : (icc ? 0 : -32768)), hi = border ? 32 : (iid ? steps : (icc ? 7 : 32767));
      const int t = h[c] < lo ? lo : (h[c] > hi ? hi : h[c]);
      bad |= t != h[c];
      h[c] = t;
    }
    rf[j] = (int32_t)xp_pack16((int16_t)h[0], (int16_t)h[1]);
  }
  return cx.wave_or(bad) != 0;
}

}  // namespace

__global__ __launch_bounds__(64 * kPsWaves, XP_WAVES_PER_EU) void xaac_ps_kernel(XaacPsParams p) {
  __shared__ XpLds sw[kPsWaves];
  __shared__ int32_t s_tabs[(kTabBytes + 3) / 4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  XpLds &s = sw[wave];
  const XsCx cx = {lane, 64};
  for (int i = threadIdx.x; i < (kTabBytes + 3) / 4; i += 64 * kPsWaves) s_tabs[i] = reinterpret_cast<const int32_t *>(&xaac_ps_tables)[i];
  __syncthreads(); /* the only workgroup barrier: from here on the waves run their own streams */
  const XpTables *tabs = reinterpret_cast<const XpTables *>(s_tabs);
#ifdef XP_STAGGER_G
  if (p.n >= 3 * (int)gridDim.x * kPsWaves)
    for (int t = 0; t < ((int)blockIdx.x * kPsWaves + wave) % XP_STAGGER_G; t++) __builtin_amdgcn_s_sleep(XP_STAGGER_SLEEP);
#endif
#ifdef XS_PROFILE
  if (threadIdx.x == 0) {
    for (int i = 0; i < 16; i++) xp_prof_acc[i] = 0;
  }
#endif
  for (int n = blockIdx.x * kPsWaves + wave; n < p.n; n += gridDim.x * kPsWaves) {
    xaac_ps_state *gps = p.state + n;
    int32_t *gx = p.x + (size_t)n * (40 * 128) + 2 * 128; /* slot 0 */
    int32_t *gr = p.xr + (size_t)n * (32 * 128);
    int16_t *par = p.par_l + 8 * (size_t)n;
    /* Everything the stream's set-up reads from global memory in flight together: the two words that say whether this is a PS
       frame at all, state, side info and the six scale parameters the core left (as a chain -- flags, then state, then
       parameters -- a stream paid three memory latencies before its first instruction of work) */
    constexpr int NS = (kHeadWords + 63) / 64, NF = (sizeof(xaac_ps_frame) / 4 + 63) / 64;
    int32_t rs[NS], rf[NF];
    const int32_t *gs = reinterpret_cast<const int32_t *>(gps), *gf = reinterpret_cast<const int32_t *>(p.frame + n);
    const int apply_v = p.sbr_frame[n].apply_processing, mode_v = p.header[n].channel_mode;
    const int par_v = lane < 6 ? par[lane] : 0;
    /* the right bank's scale and band limits, for the parameter row written at the end: read here with the rest -- read there,
       by one lane, they came back behind every store of the stream (a wave's memory operations retire in order) and the row,
       and the next stream's loads behind it, waited for that */
    const int tail_v = lane < 3 ? (lane == 0 ? gps->st_syn_scale_r : (lane == 1 ? gps->syn_lsb_r : gps->syn_usb_r)) : 0;

#pragma unroll
    for (int j = 0; j < NS; j++) rs[j] = lane + 64 * j < kHeadWords ? gs[lane + 64 * j] : 0;
#pragma unroll
    for (int j = 0; j < NF; j++) rf[j] = lane + 64 * j < (int)(sizeof(xaac_ps_frame) / 4) ? gf[lane + 64 * j] : 0;
    if (!(__builtin_amdgcn_readfirstlane(apply_v) && __builtin_amdgcn_readfirstlane(mode_v) == 3)) { /* sbr_dec.c:1246: mono this frame */
      if (lane == 0) {
        p.par_l[8 * (size_t)n + 6] = 0;
        p.par_r[8 * (size_t)n + 6] = 1;
      }
      continue;
    }
    const int lb_scale = __builtin_amdgcn_readlane(par_v, 0), ov_lb_scale = __builtin_amdgcn_readlane(par_v, 1);
    const int hb_scale = __builtin_amdgcn_readlane(par_v, 2), st_syn = __builtin_amdgcn_readlane(par_v, 3);
    const int lsb = __builtin_amdgcn_readlane(par_v, 4), usb = __builtin_amdgcn_readlane(par_v, 5);
    const int ps_scale_done = xp_init_ps_scale_regs<NS>(cx, rs, lb_scale, ov_lb_scale, hb_scale);
    const int ps_clamped = xp_frame_sanitize_regs<NF>(cx, rf); /* indices a parser cannot produce: contained, reported */
    xp_wave_sync(); /* the previous stream's state has left the LDS copy */
#pragma unroll
    for (int j = 0; j < NS; j++)
      if (lane + 64 * j < kHeadWords) reinterpret_cast<int32_t *>(&s.ps)[lane + 64 * j] = rs[j];
#pragma unroll
    for (int j = 0; j < NF; j++)
      if (lane + 64 * j < (int)(sizeof(xaac_ps_frame) / 4)) reinterpret_cast<int32_t *>(&s.pf)[lane + 64 * j] = rf[j];
    xp_wave_sync();
#ifdef XS_PROFILE
    if (threadIdx.x == 0) xp_prof_last = clock64();
#endif
    const int ps_scale =
        xp_ps_frame(cx, tabs, &s.ps, &s.pf, &s.w, gx, gr, lb_scale, ov_lb_scale, hb_scale, st_syn, lsb, usb, ps_scale_done);
    /* ---- state and the two synthesis launches' parameters ---- */
    xp_wave_sync();
    {
      int32_t *dst = reinterpret_cast<int32_t *>(gps);
      const int32_t *src = reinterpret_cast<const int32_t *>(&s.ps);
      for (int i = lane; i < kHeadWords; i += 64) dst[i] = src[i];
    }
    if (lane == 0) {
      int16_t *pr = p.par_r + 8 * (size_t)n;
      const int16_t ready = (int16_t)(st_syn - 8); /* makes the bank's own rescale a no-op: data is in place */
      par[0] = par[1] = par[2] = ready;
      pr[0] = pr[1] = pr[2] = (int16_t)ps_scale;
      pr[3] = (int16_t)__builtin_amdgcn_readlane(tail_v, 0);
      pr[4] = (int16_t)__builtin_amdgcn_readlane(tail_v, 1);
      pr[5] = (int16_t)__builtin_amdgcn_readlane(tail_v, 2);
      pr[6] = 0;
      par[6] = 0;
      gps->lb_scale_r = gps->ov_lb_scale_r = gps->hb_scale_r = (int16_t)ps_scale; /* sbr_dec.c:1261-1264 */
      p.sbr_state[n].ps_scale = (int16_t)ps_scale;
#ifndef XS_PROFILE
      if (ps_clamped && p.status) p.status[n] = -1;
#endif
    }
  }
#ifdef XS_PROFILE
  __syncthreads();
  if (threadIdx.x < 16 && p.dbg) atomicAdd(reinterpret_cast<unsigned long long *>(p.dbg) + 32 + lane, (unsigned long long)xp_prof_acc[lane]);
#endif
}

/* ixheaacd_sbrdecoder.c:762-806: what channel 1 inherits from channel 0 when a mono stream turns into a parametric-stereo
   or a stereo one.  Only the arrays and scale factors the reference copies move; channel 1 keeps its own ring positions
   (the reference's pointers into the rings are not part of the copy).  One workgroup per entry. */
__global__ __launch_bounds__(64) void xaac_sbr_handover_kernel(xaac_sbr_handover_batch b) {
  const int e = blockIdx.x, lane = threadIdx.x;
  const xaac_sbr_state *src = b.state + b.src[e];
  if (b.mode == XAAC_HANDOVER_PS_START) {
    xaac_ps_state *dst = b.ps_state + b.dst[e];
    for (int i = lane; i < 1280; i += 64) dst->syn_ring_r[i] = src->syn_ring[i];
    if (lane == 0) dst->st_syn_scale_r = src->st_syn_scale;
  } else {
    xaac_sbr_state *dst = b.state + b.dst[e];
    for (int i = lane; i < 1280; i += 64) dst->syn_ring[i] = src->syn_ring[i];
    for (int i = lane; i < 320; i += 64) dst->ana_ring[i] = src->ana_ring[i];
    for (int i = lane; i < 6 * 64; i += 64) dst->overlap[i] = src->overlap[i]; /* MAX_OV_COLS * NO_SYNTHESIS_CHANNELS words */
    if (lane == 0) {
      dst->st_syn_scale = src->st_syn_scale;
      dst->st_lb_scale = src->st_lb_scale;
      dst->ov_lb_scale = src->ov_lb_scale;
      dst->ov_hb_scale = src->ov_hb_scale;
    }
  }
}

/* sbrdecoder.c:103-252 (ixheaacd_sbr_dec_reset) and :254-276 (ixheaacd_prepare_upsamp): the state words a frame with the
   reset / up-sampling flag rewrites -- the same as libxaac_amd/host/xaac_parse.cpp: xaac_sbr_state_apply_side and
   xaac_ps_state_apply_side.  One thread per channel. */
__global__ __launch_bounds__(256) void xaac_sbr_apply_side_kernel(xaac_sbr_apply_side_batch b) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= b.n_streams * b.ch_fac) return;
  const int i = e / b.ch_fac, c = e % b.ch_fac;
  const int32_t *f = b.flags + 8 * (size_t)i;
  const int reset = f[1], reset_channels = f[2], upsampling = f[3];
  if (!reset && !upsampling) return;
  xaac_sbr_state *s = b.state + e;
  const xaac_sbr_header *h = b.header + (size_t)i * b.ch_fac; /* the stream's header (channel 0's copy, as the host one reads) */
  if (reset && c < reset_channels) {
    s->ph_index = 0;
    s->filt_buf_noise_e = 0;
    s->start_up = 1;
    s->syn_lsb = s->codec_usb = h->sub_band_start;
    s->syn_usb = h->sub_band_end;
    for (int k = 0; k < XAAC_SBR_MAX_PATCHES; k++) s->bw_array_prev[k] = 0;
  }
  if (upsampling) {
    s->syn_lsb = s->codec_usb = 32;
    s->syn_usb = 64;
  }
  if (b.ps_state && c == 0) {
    xaac_ps_state *ps = b.ps_state + i;
    if (reset && reset_channels > 1) {
      ps->syn_lsb_r = h->sub_band_start;
      ps->syn_usb_r = h->sub_band_end;
    }
    if (upsampling) {
      ps->syn_lsb_r = 32;
      ps->syn_usb_r = 64;
    }
  }
}

extern "C" hipError_t xaac_launch_sbr_apply_side(const xaac_sbr_apply_side_batch *b, hipStream_t stream) {
  const int n = b->n_streams * b->ch_fac;
  hipLaunchKernelGGL(xaac_sbr_apply_side_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, *b);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_sbr_handover(const xaac_sbr_handover_batch *b, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_sbr_handover_kernel, dim3(b->n), dim3(64), 0, stream, *b);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_ps(const XaacPsParams *p, hipStream_t stream) {
  static int resident = 0; /* workgroups the chip holds at once (LDS-bound) */
  if (!resident) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, xaac_ps_kernel, 64 * kPsWaves, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
    const char *e = getenv("XAAC_PS_WG_PER_CU"); /* developer override */
    if (e && atoi(e) > 0) per_cu = atoi(e);
    resident = per_cu * (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
  }
  const int need = (p->n + kPsWaves - 1) / kPsWaves, grid = need < resident ? need : resident;
  hipLaunchKernelGGL(xaac_ps_kernel, dim3(grid), dim3(64 * kPsWaves), 0, stream, *p);
  return hipGetLastError();
}

/* xaac_warm_up (xaac_abi.cpp): asking for a kernel's attributes puts this translation unit's code object on the device */
extern "C" hipError_t xaac_warm_sbr_ps(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&xaac_ps_kernel));
}
