/*
 * sbr_qmf_kernel.hip -- gfx950 kernels for the fixed-point SBR QMF banks:
 *   xaac_qmf_analysis_kernel   <->  ixheaacd_cplx_anal_qmffilt  (generic/ixheaacd_qmf_dec_generic.c:590)
 *   xaac_qmf_synthesis_kernel  <->  ixheaacd_cplx_synt_qmffilt  (ixheaacd_qmf_dec.c:811, no-PS branch)
 * low-power (real) and HQ (complex) modes, one frame = 32 slots per channel.
 *
 * MI355X mapping.  The reference runs the 32 slots of a frame one after another through
 * pointer-rotated ring buffers.  Both banks are time-invariant polyphase FIRs around a
 * per-slot transform, and neither window-add can saturate (sum|c| over the 5 analysis taps
 * is 32757, over the 10 synthesis taps 57308 -> |acc| < 2^31), so the accumulation order
 * is immaterial and every slot is independent given the frame's samples plus the history
 * the ring holds:
 *   analysis   z[s][m] = sum_{j<5}  x[32s+31 - (m+64j)] * c[2m+128j]
 *   synthesis  y[s][k] = rnd + sum_{A<10} v[s-A][64(A&1)+k] * c[64A+k]
 * (pairings derived by simulating the reference's pointer state machines; the ring layout
 * itself is kept as the persistent state so it stays word-identical with the reference).
 * One wave handles TWO channel-frames:
 *   - window-add phases: lanes = the 64 polyphase outputs, loop over slots, history in LDS,
 *     the lane's 5/10 coefficients held in registers;
 *   - transform phase: ONE LANE = ONE SLOT (2 x 32 slots = 64 lanes) running the scalar
 *     code of sbr_qmf.h on private registers -- the transforms are 16/32-point FFT sized,
 *     the batch supplies the parallelism, no cross-lane traffic;
 *   - rows move between the two layouts through a padded LDS tile (stride 65/129 words,
 *     conflict-free both ways), HBM sees only coalesced row accesses.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sbr_qmf.h"
#include "sbr_qmf_kernel.h"

namespace {

constexpr int kHist = 288 + 1024; /* analysis: time-ordered samples, oldest first */

/* where the analysis ring keeps the sample of age a (0 = newest) when the next write goes to wr */
__device__ __forceinline__ int ana_ring_pos(int wr, int a) {
  int p = wr + 32 + a;
  return p >= 320 ? p - 320 : p;
}

/* 32 steps of the window-phase bookkeeping of generic:698-722 (does not influence the samples) */
__device__ __forceinline__ int ana_phase_after_frame(int phase) {
  int f1 = phase, f2 = phase + 64;
  for (int s = 0; s < 32; s++) {
    f1 += 64;
    f2 += 64;
    int t = f1;
    f1 = f2;
    f2 = t;
    if (f2 > 640) {
      f1 = 0;
      f2 = 64;
    }
  }
  return f1;
}

/* env_calc.c:1099 */
__device__ __forceinline__ int32_t adj_scale(int32_t v, int shift) {
  if (shift > 31) shift = 31;
  if (shift < -31) shift = -31;
  return shift > 0 ? fx_shlw(v, shift) : (shift < 0 ? (v >> -shift) : v);
}

}  // namespace

/* ===================================================================================== */
template <bool LP>
__global__ __launch_bounds__(XAAC_QMF_BLOCK) void xaac_qmf_analysis_kernel(XaacQmfAnaParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char *wbase = smem + wave * XAAC_QMF_ANA_LDS_PER_WAVE;
  int16_t *hist = reinterpret_cast<int16_t *>(wbase);            /* [2][kHist] */
  int32_t *z = reinterpret_cast<int32_t *>(wbase + 2 * kHist * 2 + 32); /* [64][65] */

  int32_t coef[5]; /* c[2m + 128 j] of this lane's polyphase branch m = lane */
#pragma unroll
  for (int j = 0; j < 5; j++) coef[j] = xaac_qmf_qmf_c[2 * lane + 128 * j];

  if (p.zero_words && blockIdx.x == 0 && threadIdx.x < 2) p.zero_words[threadIdx.x] = 0;
  const int n_pairs = (p.n_ch + 1) >> 1;
  const int waves_total = gridDim.x * XAAC_QMF_WAVES;
  for (int pair = blockIdx.x * XAAC_QMF_WAVES + wave; pair < n_pairs; pair += waves_total) {
    /* ---- history + new samples, time ordered ------------------------------------------ */
    int wr_v[2] = {0, 0}, ph_v[2] = {0, 0};
    {
      /* all loads of both channels (5 ring + 16 PCM per lane and channel) in flight before the first LDS store */
      /* (the rings by position, not by age: their loads then do not wait for the write positions, which only say where in the
         time-ordered history a fetched sample belongs; write position and window phase are read here for the state update too) */
      int16_t hr[2][5], hp[2][16];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int ch = 2 * pair + c;
        if (ch < p.n_ch) { /* (uniform) */
          const xaac_qmf_ana_state *st = reinterpret_cast<const xaac_qmf_ana_state *>(
              reinterpret_cast<const char *>(p.state) + (size_t)ch * p.state_stride);
          wr_v[c] = st->wr;
          ph_v[c] = st->phase;
          const int cf = p.ch_fac;
          const int16_t *src = p.pcm + (size_t)(ch / cf) * 1024 * cf + (ch % cf);
#pragma unroll
          for (int j = 0; j < 5; j++) hr[c][j] = st->ring[lane + 64 * j];
#pragma unroll
          for (int j = 0; j < 16; j++) hp[c][j] = src[(size_t)(lane + 64 * j) * cf];
        }
      }
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int ch = 2 * pair + c;
        int16_t *h = hist + c * kHist;
        if (ch < p.n_ch) {
          const int wr = __builtin_amdgcn_readfirstlane(wr_v[c]);
#pragma unroll
          for (int j = 0; j < 5; j++) {
            int a = lane + 64 * j - wr - 32; /* age of the sample at this position: ana_ring_pos(wr, a) = position */
            a += a < 0 ? 320 : 0;
            a += a < 0 ? 320 : 0;
            if (a < 288) h[287 - a] = hr[c][j];
          }
#pragma unroll
          for (int j = 0; j < 16; j++) h[288 + lane + 64 * j] = hp[c][j];
        } else {
          for (int i = lane; i < kHist; i += 64) h[i] = 0;
        }
      }
    }
    /* ---- window-add: lanes = polyphase branch m; a channel's 40 history words of the branch are read once and the 32 slots'
       sums formed from registers (as a loop over the slots with five 2-byte LDS reads each, and scalar address arithmetic
       around them, this was most of the low-power bank's time: 1.6 scalar instructions per vector one) ------------- */
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int16_t *h = hist + c * kHist + 288 + 31 - lane;
      int32_t u[40]; /* u[8 + k] = x[32 k + 31 - m], k = -8 .. 31 */
#pragma unroll
      for (int k = 0; k < 40; k++) u[k] = h[32 * (k - 8)];
#pragma unroll
      for (int sl = 0; sl < 32; sl++) {
        int32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) acc += u[8 + sl - 2 * j] * coef[j]; /* |acc| < 2^30: exact */
        z[65 * (32 * c + sl) + lane] = acc;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    /* ---- per-slot transform: lane = slot ------------------------------------------------ */
    {
      int32_t in[64];
#pragma unroll
      for (int k = 0; k < 64; k++) in[k] = z[65 * lane + k];
      if (LP) {
        int32_t out[32];
        xq_dct3_32(in, out);
#pragma unroll
        for (int k = 0; k < 32; k++) z[65 * lane + k] = out[k];
      } else {
        int32_t sb[128], t[128];
        int nrot = p.usb;
        if (p.frame) { /* lanes 0..31 are the slots of the pair's first channel, 32..63 of its second */
          const int ch = 2 * pair + (lane >> 5);
          if (ch < p.n_ch) {
            const xaac_sbr_frame *f = p.frame + ch;
            const xaac_sbr_state *sst = reinterpret_cast<const xaac_sbr_state *>(
                reinterpret_cast<const char *>(p.state) + (size_t)ch * p.state_stride);
            nrot = f->apply_processing ? f->max_qmf_subband_aac : sst->codec_usb;
          }
        }
        xq_fwd_modulation(in, sb, t, nrot);
#pragma unroll
        for (int k = 0; k < 32; k++) {
          z[65 * lane + k] = sb[k];
          z[65 * lane + 32 + k] = sb[64 + k];
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    /* ---- rows out: coalesced, real bands at +0, imaginary (HQ) at +64 ---------------------- */
    if (LP) {
      /* 32 real bands per row: lanes 0..31 write the first channel's row r, lanes 32..63 the second channel's */
      const int ch = 2 * pair + (lane >> 5);
      if (ch < p.n_ch) {
        int32_t *row = p.qmf + (size_t)ch * p.qmf_ch_stride + (lane & 31);
        const int32_t *src = z + 65 * 32 * (lane >> 5) + (lane & 31);
#pragma unroll 8
        for (int r = 0; r < 32; r++) row[(size_t)r * p.slot_stride] = src[65 * r];
      }
    } else {
      for (int r = 0; r < 64; r++) {
        const int ch = 2 * pair + (r >> 5);
        if (ch >= p.n_ch) break;
        int32_t *row = p.qmf + (size_t)ch * p.qmf_ch_stride + (size_t)(r & 31) * p.slot_stride;
        row[(lane & 31) + 64 * (lane >> 5)] = z[65 * r + lane];
      }
    }
    /* ---- state: the ring as the reference leaves it after 32 slots -------------------------- */
    for (int c = 0; c < 2; c++) {
      const int ch = 2 * pair + c;
      if (ch >= p.n_ch) break;
      xaac_qmf_ana_state *st =
          reinterpret_cast<xaac_qmf_ana_state *>(reinterpret_cast<char *>(p.state) + (size_t)ch * p.state_stride);
      const int wr_new = (__builtin_amdgcn_readfirstlane(wr_v[c]) + 256) % 320;
      const int ph_new = ana_phase_after_frame(__builtin_amdgcn_readfirstlane(ph_v[c]));
      const int16_t *h = hist + c * kHist;
      for (int a = lane; a < 320; a += 64) st->ring[ana_ring_pos(wr_new, a)] = h[kHist - 1 - a];
      if (lane == 0) {
        st->wr = (int16_t)wr_new;
        st->phase = (int16_t)ph_new;
      }
    }
  }
}

/* ===================================================================================== */
/* The complex (HQ) analysis bank, two channels per 128-thread workgroup.  Same arithmetic as xaac_qmf_analysis_kernel<false>
   (sbr_qmf.h: xq_fwd_modulation), arranged like the synthesis pair kernel below:
   - wave w window-adds channel w: lane = polyphase branch m, and since z[s][m] = sum_j u[s - 2j] c[2m + 128j] with
     u[k] = x[32k + 31 - m], the lane keeps its forty u in registers -- one LDS read per slot instead of five;
   - the slot transform's two independent halves (xq_cos_sin_mod_half<16, H>) run on the two waves, lane = (channel,
     slot) on both: wave 0 takes the difference terms (real parts), wave 1 the sums (imaginary parts), 32 words per
     lane instead of the 64 + 128 + 128 of the whole transform; they meet through two 64 x 33-word tiles for the final
     rotation (wave 0 forms the real outputs, wave 1 the imaginary ones) and for the way out (lane = band);
   - 22 KB of LDS per workgroup: seven workgroups (14 waves) per CU where the one-wave-per-pair kernel had six waves. */
__global__ __launch_bounds__(128) void xaac_qmf_analysis_hq_kernel(XaacQmfAnaParams p) {
  __shared__ int16_t hist[2][kHist];
  __shared__ int32_t tile[2 * 64 * 33]; /* the 64 x 65 window-add tile, then the two 64 x 33 half tiles */
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.x;
  const int chw = 2 * pair + w; /* the channel this wave loads, window-adds and writes */
  const bool live = chw < p.n_ch;
  if (p.zero_words && blockIdx.x == 0 && threadIdx.x < 2) p.zero_words[threadIdx.x] = 0;
  xaac_qmf_ana_state *st = reinterpret_cast<xaac_qmf_ana_state *>(reinterpret_cast<char *>(p.state) + (size_t)(live ? chw : 0) * p.state_stride);
  int wr = 0, phase_old = 0;
  /* what the final rotation needs of the lane's channel (lanes 0..31: the pair's first channel, 32..63: its second): read
     here, with everything else the workgroup reads, instead of as two dependent round trips in front of the rotation */
  int rot_apply = 0, rot_sub = 0, rot_usb = 0;
  const bool rot_live = p.frame && 2 * pair + (lane >> 5) < p.n_ch;
  if (rot_live) {
    const int ch = 2 * pair + (lane >> 5);
    const xaac_sbr_frame *f = p.frame + ch;
    const xaac_sbr_state *sst = reinterpret_cast<const xaac_sbr_state *>(reinterpret_cast<const char *>(p.state) + (size_t)ch * p.state_stride);
    rot_apply = f->apply_processing;
    rot_sub = f->max_qmf_subband_aac;
    rot_usb = sst->codec_usb;
  }
  /* ---- history + new samples of channel w, time ordered ---- */
  {
    int16_t *h = hist[w];
    if (live) {
      /* the ring is fetched by position, not by age: the loads then do not wait for the write position, which only says
         where in the time-ordered history a fetched sample belongs */
      const int wr_v = st->wr;
      phase_old = st->phase;
      const int cf = p.ch_fac;
      const int16_t *src = p.pcm + (size_t)(chw / cf) * 1024 * cf + (chw % cf);
      int16_t hr[5], hp[16];
#pragma unroll
      for (int j = 0; j < 5; j++) hr[j] = st->ring[lane + 64 * j];
#pragma unroll
      for (int j = 0; j < 16; j++) hp[j] = src[(size_t)(lane + 64 * j) * cf];
      wr = __builtin_amdgcn_readfirstlane(wr_v);
#pragma unroll
      for (int j = 0; j < 5; j++) {
        int a = lane + 64 * j - wr - 32; /* age of the sample at this position: ana_ring_pos(wr, a) = position */
        a += a < 0 ? 320 : 0;
        a += a < 0 ? 320 : 0;
        if (a < 288) h[287 - a] = hr[j];
      }
#pragma unroll
      for (int j = 0; j < 16; j++) h[288 + lane + 64 * j] = hp[j];
    } else {
      for (int i = lane; i < kHist; i += 64) h[i] = 0;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* hist[w] is this wave's own */
  /* ---- window-add of channel w: lane = polyphase branch m ---- */
  {
    int32_t coef[5];
#pragma unroll
    for (int j = 0; j < 5; j++) coef[j] = xaac_qmf_qmf_c[2 * lane + 128 * j];
    const int16_t *h = hist[w] + 288 + 31 - lane;
    int32_t u[40]; /* u[8 + k] = x[32 k + 31 - m], k = -8 .. 31 */
#pragma unroll
    for (int k = 0; k < 40; k++) u[k] = h[32 * (k - 8)];
#pragma unroll
    for (int sl = 0; sl < 32; sl++) {
      int32_t acc = 0;
#pragma unroll
      for (int j = 0; j < 5; j++) acc += u[8 + sl - 2 * j] * coef[j]; /* |acc| < 2^30: exact */
      tile[65 * (32 * w + sl) + lane] = acc;
    }
  }
  __syncthreads();
  /* ---- the halves: lane = (channel, slot) ---- */
  int32_t s[32];
  {
    int32_t in[64];
#pragma unroll
    for (int k = 0; k < 64; k++) in[k] = tile[65 * lane + k];
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const int32_t a = fx_shr(in[i], 4), b = fx_shr(in[63 - i], 4);
      s[i] = w == 0 ? fx_sub_sat(a, b) : fx_add_sat(a, b);
    }
  }
  __syncthreads(); /* the window-add tile is dead */
  {
    int32_t t[32];
    if (w == 0)
      xq_cos_sin_mod_half<16, 0>(s, t);
    else
      xq_cos_sin_mod_half<16, 1>(s, t);
  }
  int32_t *mine = tile + w * (64 * 33);
  const int32_t *other = tile + (1 - w) * (64 * 33);
#pragma unroll
  for (int k = 0; k < 32; k++) mine[33 * lane + k] = s[k];
  __syncthreads();
  {
    int32_t o[32];
#pragma unroll
    for (int k = 0; k < 32; k++) o[k] = other[33 * lane + k];
    /* lanes 0..31 are the slots of the pair's first channel, 32..63 of its second */
    const int nrot = rot_live ? (rot_apply ? rot_sub : rot_usb) : p.usb;
    const int16_t *tc = XQ_T(t_cos_sin_l32);
#pragma unroll
    for (int i = 0; i < 32; i++) { /* generic:650-656: own = real (wave 0) / imaginary (wave 1) part, o = the other */
      const int16_t c = tc[2 * i], sn = tc[2 * i + 1];
      const int32_t pa = fx_mul32x16_shl(s[i], c), pb = fx_mul32x16_shl(o[i], sn);
      const int32_t r = w == 0 ? fx_add_sat(pa, pb) : fx_sub_sat(pa, pb);
      s[i] = i < nrot ? r : s[i];
    }
  }
  __syncthreads(); /* both waves have read the other's half */
#pragma unroll
  for (int k = 0; k < 32; k++) mine[33 * lane + k] = s[k];
  __syncthreads();
  if (!live) return;
  /* ---- channel w's rows out: lane = (part, band), real bands at +0, imaginary at +64 ---- */
  {
    int32_t *row = p.qmf + (size_t)chw * p.qmf_ch_stride + (lane & 31) + 64 * (lane >> 5);
    const int32_t *src = tile + (lane >> 5) * (64 * 33) + 33 * (32 * w) + (lane & 31);
#pragma unroll 8
    for (int r = 0; r < 32; r++) row[(size_t)r * p.slot_stride] = src[33 * r];
  }
  /* ---- state: the ring as the reference leaves it after 32 slots ---- */
  {
    const int wr_new = (wr + 256) % 320;
    const int ph_new = ana_phase_after_frame(phase_old);
    const int16_t *h = hist[w];
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int a = lane + 64 * j;
      st->ring[ana_ring_pos(wr_new, a)] = h[kHist - 1 - a];
    }
    if (lane == 0) {
      st->wr = (int16_t)wr_new;
      st->phase = (int16_t)ph_new;
    }
  }
}

#ifdef XS_PROFILE
/* phase timers of the synthesis kernel (tools/prof_sbr_core.py): cycles of each wave's lane 0 */

#define XQ_TIME(i)                                                            \
  do {                                                                     \
    if (lane == 0) {                                                       \
      long long t_ = clock64();                                            \
      xq_acc[i] += t_ - xq_last;                                           \
      xq_last = t_;                                                        \
    }                                                                      \
  } while (0)
#else

#define XQ_TIME(i)
#endif

/* ===================================================================================== */
/* DS: the down-sampled bank -- 32 channels: the same kernel with half-size slot blocks (NC) */
template <bool LP, bool DS>
__global__ __launch_bounds__(XAAC_QMF_BLOCK) void xaac_qmf_synthesis_kernel(XaacQmfSynParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROW = LP ? 64 : 128;    /* words per slot row */
  constexpr int RS = ROW + 1;           /* padded LDS row stride */
  constexpr int VSLOTS = 9 + 32;        /* 9 slots of history + this frame */
  constexpr int NC = DS ? 32 : 64;      /* synthesis channels = output samples per slot */
  constexpr int BLK = 2 * NC;           /* ring samples one slot adds */
  constexpr int RING = 10 * BLK;
  constexpr int VROW = BLK + 2;         /* LDS stride of a slot's ring samples: odd in dwords, so that the 64 lanes
                                           writing their slots (lane = slot) land in different banks */
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char *wbase = smem + wave * (LP ? XAAC_QMF_SYN_LDS_PER_WAVE_LP : XAAC_QMF_SYN_LDS_PER_WAVE_HQ);
  int32_t *rows = reinterpret_cast<int32_t *>(wbase);  /* [64][RS] slot rows, later aliased by ... */
  int16_t *v = reinterpret_cast<int16_t *>(wbase);     /* ... [2][VSLOTS][VROW] ring samples */

  int32_t coef[10]; /* c[64 A + k], k = lane; down-sampled: every second one, k = lane & 31 (qmf_dec.c:749) */
#pragma unroll
  for (int a = 0; a < 10; a++) coef[a] = xaac_qmf_qmf_c[64 * a + (DS ? 2 * (lane & 31) : lane)];

#ifdef XS_PROFILE
  long long xq_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xq_last = clock64();
#endif
  const int n_pairs = (p.n_ch + 1) >> 1;
  const int waves_total = gridDim.x * XAAC_QMF_WAVES;
  for (int pair = blockIdx.x * XAAC_QMF_WAVES + wave; pair < n_pairs; pair += waves_total) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    XQ_TIME(0);
    /* ---- slot rows in, coalesced --------------------------------------------------------------- */
    /* a group of rows' loads is issued before the first is consumed: one HBM/L2 latency per group, not per row */
#ifndef XQ_SYN_GROUP_LP
#define XQ_SYN_GROUP_LP 16 /* measured on the LP chain: 8: 141 us, 16: 129, 32: 129, 64: 132 */
#endif
    constexpr int G = LP ? XQ_SYN_GROUP_LP : 8;
    /* the lane's channel's scale row, the two ring offsets and the two inactive flags: asked for here, in front of the rows, and
       looked at behind the transform -- read where they are used they were three more memory round trips per pair */
    int16_t sfv[6] = {0, 0, 0, 0, 0, 0};
    int d_v[2] = {0, 0}, off_v[2] = {0, 0};
    {
      const int ch = 2 * pair + (lane >> 5);
      const int16_t *sf = p.scale + (size_t)p.scale_stride * (ch < p.n_ch ? ch : p.n_ch - 1);
#pragma unroll
      for (int k = 0; k < 4; k++) sfv[k] = sf[k];
      if (p.per_ch_bands) {
        sfv[4] = sf[4];
        sfv[5] = sf[5];
      }
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int chc = 2 * pair + c;
        if (chc < p.n_ch) {
          d_v[c] = reinterpret_cast<const xaac_qmf_syn_state *>(reinterpret_cast<const char *>(p.state) + (size_t)chc * p.state_stride)->drc_offset;
          if (p.per_ch_bands) off_v[c] = p.scale[(size_t)p.scale_stride * chc + 6];
        }
      }
    }
    /* Low-power banks: the 2 x 9 history slots of the ring are asked for while the rows still arrive -- as pairs of samples (4-byte
       words; the ring's offset is even unless a caller hands in an odd one), kept in registers through the transform and
       put into place behind it: read there, behind the transform, their loads paid a memory latency of their own */
    constexpr bool EARLY = LP;
    constexpr int HP = (9 * BLK / 2 + 63) / 64; /* words per lane and channel */
    int32_t hpair[2][HP];
    bool hist_words = false;
    for (int r0 = 0; r0 < 64; r0 += G) {
      int32_t tmp[G][ROW / 64];
#pragma unroll
      for (int j = 0; j < G; j++) {
        const int r = r0 + j, ch = 2 * pair + (r >> 5);
        const int32_t *row = p.qmf + (size_t)(ch < p.n_ch ? ch : 0) * p.qmf_ch_stride + (size_t)(r & 31) * p.slot_stride;
#pragma unroll
        for (int q = 0; q < ROW / 64; q++) tmp[j][q] = row[lane + 64 * q];
      }
#pragma unroll
      for (int j = 0; j < G; j++) {
        const int r = r0 + j, ch = 2 * pair + (r >> 5);
#pragma unroll
        for (int q = 0; q < ROW / 64; q++) rows[RS * r + lane + 64 * q] = ch < p.n_ch ? tmp[j][q] : 0;
      }
      if (EARLY && r0 == 0) { /* the ring offsets have arrived with the first rows */
        const int d0 = __builtin_amdgcn_readfirstlane(d_v[0]), d1 = __builtin_amdgcn_readfirstlane(d_v[1]);
        hist_words = ((d0 | d1) & 1) == 0 && (p.state_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(p.state) & 3) == 0;
        if (hist_words) {
#pragma unroll
          for (int c = 0; c < 2; c++) {
            const int chc = 2 * pair + c;
            const xaac_qmf_syn_state *st = reinterpret_cast<const xaac_qmf_syn_state *>(
                reinterpret_cast<const char *>(p.state) + (size_t)(chc < p.n_ch ? chc : 0) * p.state_stride);
            const int d = c ? d1 : d0;
#pragma unroll
            for (int j = 0; j < HP; j++) {
              int i = 2 * (lane + 64 * j);
              i = i < 9 * BLK ? i : 0;
              const int A = 9 - i / BLK;
              int pos = d + BLK * A + i % BLK;
              if (pos >= RING) pos -= RING;
              hpair[c][j] = *reinterpret_cast<const int32_t *>(&st->ring[pos]);
            }
          }
        }
      }
    }
    XQ_TIME(1);
    /* ---- per-slot: region rescale (qmf_dec.c:937-953) + inverse modulation; lane = slot ----------- */
    int16_t b[BLK];
    {
      const int16_t *sf = sfv; /* lb_scale, ov_lb_scale, hb_scale, st_syn_scale (+ lsb, usb) of the lane's channel */
      const int st_syn = sf[3];
      const int lsb = p.per_ch_bands ? sf[4] : p.lsb, usb = p.per_ch_bands ? sf[5] : p.usb;
      const int bias = LP ? 4 : 8;
      const int s = lane & 31;
      const int lo_shift = (st_syn - (s < p.split ? sf[1] : sf[0])) - bias;
      const int hb_shift = (st_syn - sf[2]) - bias;
      int32_t x[ROW], t[ROW];
#pragma unroll
      for (int k = 0; k < ROW; k++) {
        const int band = k & 63;
        if (band < NC) { /* the down-sampled bank transforms bands 0..31 only */
          int32_t val = rows[RS * lane + k];
          val = band < lsb ? adj_scale(val, lo_shift) : (band < usb ? adj_scale(val, hb_shift) : val);
          x[k] = val;
        }
      }
      if (LP) {
        if (DS)
          xq_dct2_32_lp(x, t, b);
        else
          xq_dct2_64_lp(x, t, b);
      } else {
        if (DS)
          xq_synth_hq_slot_ds(x, t, b, -(st_syn - 3) + 1);
        else
          xq_synth_hq_slot(x, t, b, -(st_syn - 3) + 1);
      }
    }
    XQ_TIME(2);
    /* all lanes hold their slot in registers now: the row tile may be overwritten (the tile is
       re-used through an int16 view: keep the compiler from moving accesses across this point) */
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      int16_t *dst = v + ((lane >> 5) * VSLOTS + 9 + (lane & 31)) * VROW;
#pragma unroll
      for (int i = 0; i < BLK; i += 2)
        *reinterpret_cast<int32_t *>(dst + i) = (int32_t)((uint32_t)(uint16_t)b[i] | ((uint32_t)(uint16_t)b[i + 1] << 16));
    }
    if (EARLY && hist_words) {
#pragma unroll
      for (int c = 0; c < 2; c++) {
        if (2 * pair + c >= p.n_ch) break;
#pragma unroll
        for (int j = 0; j < HP; j++) {
          const int i = 2 * (lane + 64 * j);
          if (i < 9 * BLK) *reinterpret_cast<int32_t *>(&v[(c * VSLOTS + i / BLK) * VROW + i % BLK]) = hpair[c][j];
        }
      }
    }
    for (int c = 0; c < 2; c++) {
      const int ch = 2 * pair + c;
      if (ch >= p.n_ch) break;
      if (EARLY && hist_words) break; /* (in place already) */
      const xaac_qmf_syn_state *st = reinterpret_cast<const xaac_qmf_syn_state *>(
          reinterpret_cast<const char *>(p.state) + (size_t)ch * p.state_stride);
      const int d = __builtin_amdgcn_readfirstlane(d_v[c]);
      constexpr int HJ = 9 * BLK / 64;
      int16_t hist[HJ]; /* 9 slots x BLK samples over 64 lanes: all loads in flight together */
#pragma unroll
      for (int j = 0; j < HJ; j++) {
        const int i = lane + 64 * j;
        const int A = 9 - i / BLK; /* slot age relative to this frame's slot 0 */
        int pos = d + BLK * A + i % BLK;
        if (pos >= RING) pos -= RING;
        hist[j] = st->ring[pos];
      }
#pragma unroll
      for (int j = 0; j < HJ; j++) {
        const int i = lane + 64 * j;
        v[(c * VSLOTS + i / BLK) * VROW + i % BLK] = hist[j];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    XQ_TIME(3);
    /* ---- window-add: lanes = output sample k of the slot, loop over slots ------------------------ */
    for (int c = 0; c < 2; c++) {
      const int ch = 2 * pair + c;
      if (ch >= p.n_ch) break;
      if (p.per_ch_bands && __builtin_amdgcn_readfirstlane(off_v[c])) continue; /* channel inactive this frame */
      const int cf = p.pcm_sample_stride ? p.pcm_sample_stride : p.ch_fac;
      int16_t *dst = p.pcm_sample_stride ? p.pcm + (size_t)ch * p.pcm_ch_stride
                                         : p.pcm + (size_t)(ch / cf) * (32 * NC) * cf + (ch % cf);
      const int shift = LP ? 2 : 1;
      /* a wave covers one slot of 64 samples, or two slots of 32 */
      for (int s0 = 0; s0 < 32; s0 += 64 / NC) {
        const int s = DS ? s0 + (lane >> 5) : s0, k = DS ? (lane & 31) : lane;
        const int16_t *vs = v + (c * VSLOTS + 9 + s) * VROW + k;
        int32_t acc = 0x8000 >> shift;
#pragma unroll
        for (int A = 0; A < 10; A++) acc += (int32_t)vs[-VROW * A + NC * (A & 1)] * coef[A]; /* < 2^31: exact */
        dst[(size_t)(NC * s + k) * cf] = (int16_t)(fx_shl_sat(acc, shift) >> 16);
      }
    }
    XQ_TIME(4);
    /* ---- state: ring blocks of the last 10 slots, drc offset, window phase --------------------------- */
    for (int c = 0; c < 2; c++) {
      const int ch = 2 * pair + c;
      if (ch >= p.n_ch) break;
      if (p.per_ch_bands && __builtin_amdgcn_readfirstlane(off_v[c])) continue;
      xaac_qmf_syn_state *st =
          reinterpret_cast<xaac_qmf_syn_state *>(reinterpret_cast<char *>(p.state) + (size_t)ch * p.state_stride);
      const int d_new = (__builtin_amdgcn_readfirstlane(d_v[c]) + RING - (32 * BLK) % RING) % RING; /* 32 slots of BLK downwards */
      const int ph_new = (st->phase + 128) % 640;
      for (int i = lane; i < RING; i += 64) {
        const int A = 1 + i / BLK; /* age relative to the NEXT frame's slot 0: 1..10 */
        int pos = d_new + BLK * A + i % BLK;
        if (pos >= RING) pos -= RING;
        if (pos >= RING) pos -= RING;
        st->ring[pos] = v[(c * VSLOTS + 9 + 32 - A) * VROW + i % BLK];
      }
      if (lane == 0) {
        st->drc_offset = (int16_t)d_new;
        st->phase = (int16_t)ph_new;
      }
    }
    XQ_TIME(5);
  }
#ifdef XS_PROFILE
  if (lane == 0 && p.dbg)
    for (int i = 0; i < 8; i++)
      atomicAdd(reinterpret_cast<unsigned long long *>(p.dbg) + 64 + (LP ? 0 : 8) + i, (unsigned long long)xq_acc[i]);
#endif
}

/* ===================================================================================== */
/* HE-AACv2: the left and the right complex synthesis bank of one stream in one 128-thread workgroup
   (xaac_qmf_synthesis_kernel<false, false> twice; the arithmetic is sbr_qmf.h's, only the arrangement differs):
   - the slot transform's two independent halves (sbr_qmf.h: xq_cos_sin_mod_half) run on the workgroup's two waves:
     wave h takes words 64 h .. 64 h + 63 of all 64 rows (2 channels x 32 slots, lane = (channel, slot)) through its own
     64 x 65-word LDS tile and transforms them with H = h.  A wave's code is uniform, a lane holds 2 x 64 words (no
     spill; the one-wave version kept three 64-word arrays and spilled 82 registers), and the halves meet through the
     tiles: wave 0 forms the ring samples b[0..63] of every slot, wave 1 b[64..127];
   - the region rescale (qmf_dec.c:937-953) happens on the way into the tile, where lane = band: the shift of a band is
     a lane constant per channel and slot range;
   - ring samples are stored as the pairs the window-add consumes: E[s][k] = (v[s][k], v[s-1][64+k]), so that
       y[s][k] = rnd + sum_{j<5} v[s-2j][k] c[128j+k] + v[s-2j-1][64+k] c[128j+64+k]
     is five v_dot2_i32_i16 on five ds_read_b32 (the sums cannot overflow: sum |c| = 57308, see the file header).  Each
     wave window-adds 16 slots of both channels, lane = sample, and stores interleaved L,R words;
   - the ring state keeps the reference's layout (2-byte accesses in and out).
   A state whose drc_offset is not one of the reference's (a multiple of 128 below 1280) is refused: status -1, nothing
   written for that stream. */
namespace {
__device__ __forceinline__ int32_t pair_rescale(int32_t v, int shl, int shr) { return (int32_t)((uint32_t)v << shl) >> shr; }
/* clamp(a, lo, hi) with lo <= hi as ONE v_med3_i32 (the compiler, which cannot know lo <= hi, spends a max, a compare
   and a select on the two-sided form: 256 of the kernel's 3700 vector instructions per wave) */
__device__ __forceinline__ int32_t clamp_med3(int32_t a, int32_t lo, int32_t hi) {
  int32_t r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(lo), "v"(hi));
  return r;
}
typedef short xq_short2 __attribute__((ext_vector_type(2)));
}  // namespace

#ifndef XQ_STAGGER_G
#define XQ_STAGGER_G 64
#define XQ_STAGGER_SLEEP 4    /* x 64 cycles */
#endif
#ifndef XQ_STAGGER_FIRST
#define XQ_STAGGER_FIRST 1536 /* six workgroups per CU: the ones that start together */
#endif
__global__ __launch_bounds__(128) void xaac_qmf_synthesis_pair_kernel(XaacQmfSynPairParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = 65;          /* padded LDS row stride (dwords) of tiles and pair rows */
  constexpr int EROWS = 9 + 32 + 1; /* pair rows of a channel: slots -9 .. 32 */
  constexpr int RING = 1280;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  /* a wave's tile holds the 32 half rows of ONE channel at a time (the lanes of channel 0, then those of channel 1, go
     through it): 2 x 8.3 KB instead of 2 x 16.6 KB, so that the pair rows (21.8 KB) set the workgroup's LDS and six
     workgroups -- three waves per SIMD, what the 154 VGPRs allow -- share a CU instead of four */
  int32_t *tile_own = reinterpret_cast<int32_t *>(smem) + w * 32 * RS;
  const int32_t *tile_oth = reinterpret_cast<const int32_t *>(smem) + (1 - w) * 32 * RS;
  int32_t *E = reinterpret_cast<int32_t *>(smem); /* [2][EROWS][RS], aliases the tiles once they are dead */
  const int lch = lane >> 5, lrow = lane & 31; /* the lane's (channel, slot) */
  const int i = blockIdx.x;
  /* staggered start, as in the SBR core kernel (sbr_core_kernel.hip): the workgroups the chip takes at once begin with 32 KB
     of row loads each; spread over 6.6 us (64 groups x 0.1 us) their bursts do not meet.  112.5 -> 109.3 us. */
  if (XQ_STAGGER_G > 1 && i < XQ_STAGGER_FIRST && (int)gridDim.x >= 3 * XQ_STAGGER_FIRST)
    for (int t = 0; t < i % XQ_STAGGER_G; t++) __builtin_amdgcn_s_sleep(XQ_STAGGER_SLEEP);
  /* ---- phase A: half rows in (lane = band), rescaled, through the tile to lane = (channel, slot) ---------------
     Everything the workgroup's head reads is in flight together: the two channels' parameter rows (eight shorts each: scales,
     band limits, the inactive flag), the two ring offsets and the 64 half rows.  (Read where they
     were used -- a scale inside the branch that needed it, the offsets in front of the validity check, the rows behind
     all of them -- a workgroup paid five or six memory round trips before its first row arrived.) */
  typedef int xq_int4 __attribute__((ext_vector_type(4)));
  xaac_qmf_syn_state *st_w = reinterpret_cast<xaac_qmf_syn_state *>(reinterpret_cast<char *>(p.state[w]) + (size_t)i * p.state_stride[w]);
  const xaac_qmf_syn_state *st_o = reinterpret_cast<const xaac_qmf_syn_state *>(reinterpret_cast<const char *>(p.state[1 - w]) + (size_t)i * p.state_stride[1 - w]);
  const xq_int4 par0 = *reinterpret_cast<const xq_int4 *>(p.scale[0] + 8 * (size_t)i);
  const xq_int4 par1 = *reinterpret_cast<const xq_int4 *>(p.scale[1] + 8 * (size_t)i);
  const int d_old_v = st_w->drc_offset, d_oth_v = st_o->drc_offset;
  int32_t x[64];
  int32_t tmp[64]; /* (behind the small loads: memory operations return in order, so what follows waits for those four only) */
#pragma unroll
  for (int r = 0; r < 64; r++) {
    const int c = r >> 5;
    tmp[r] = (p.qmf[c] + (size_t)i * p.qmf_stride[c] + (size_t)(r & 31) * 128 + 64 * w)[lane];
  }
  const auto lo16 = [](int v) { return (int)(int16_t)v; };
  const auto hi16 = [](int v) { return v >> 16; };
  const int u0 = __builtin_amdgcn_readfirstlane(par0.x), u1 = __builtin_amdgcn_readfirstlane(par0.y);
  const int u2 = __builtin_amdgcn_readfirstlane(par0.z), u3 = __builtin_amdgcn_readfirstlane(par0.w);
  const int v0 = __builtin_amdgcn_readfirstlane(par1.x), v1 = __builtin_amdgcn_readfirstlane(par1.y);
  const int v2 = __builtin_amdgcn_readfirstlane(par1.z), v3 = __builtin_amdgcn_readfirstlane(par1.w);
  const int inactive0 = lo16(u3), inactive1 = lo16(v3);
  const int st_syn0 = hi16(u1), st_syn1 = hi16(v1);
  /* this wave's channel for history / state: channel w */
  const int d_old = __builtin_amdgcn_readfirstlane(d_old_v);
  {
    const int d_oth = __builtin_amdgcn_readfirstlane(d_oth_v);
    if (((d_old | d_oth) & 127) != 0 || d_old < 0 || d_old >= RING || d_oth < 0 || d_oth >= RING) {
      if (threadIdx.x == 0 && p.status) p.status[i] = -1;
      return; /* uniform over the workgroup */
    }
  }
  {
    int shl[2][2], shr[2][2]; /* [channel][slot < split] for band = 64 w' + lane -> the band is lane (both halves: re | im of band lane) */
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int w0 = c ? v0 : u0, w1 = c ? v1 : u1, w2 = c ? v2 : u2;
      const int st_syn = hi16(w1), lsb = lo16(w2), usb = hi16(w2), hb = lo16(w1);
#pragma unroll
      for (int ov = 0; ov < 2; ov++) {
        const int lo_sf = ov ? hi16(w0) : lo16(w0); /* sf[ov]: lb_scale, ov_lb_scale */
        int sh = lane < lsb ? (st_syn - lo_sf) - 8 : (lane < usb ? (st_syn - hb) - 8 : 0);
        sh = sh > 31 ? 31 : (sh < -31 ? -31 : sh); /* env_calc.c:1099 */
        shl[c][ov] = sh > 0 ? sh : 0;
        shr[c][ov] = sh < 0 ? -sh : 0;
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
#pragma unroll
      for (int j = 0; j < 32; j++) {
        const bool ov = j < p.split;
        tile_own[RS * j + lane] = pair_rescale(tmp[32 * c + j], ov ? shl[c][1] : shl[c][0], ov ? shr[c][1] : shr[c][0]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* the tile is this wave's own: no barrier */
      if (lch == c) {
#pragma unroll
        for (int k = 0; k < 64; k++) x[k] = tile_own[RS * lrow + k];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* channel 0's reads are done before channel 1's rows land */
    }
  }
  {
    int32_t t[64];
    if (w == 0)
      xq_cos_sin_mod_half<32, 0>(x, t);
    else
      xq_cos_sin_mod_half<32, 1>(x, t);
  }
  /* ---- phase B: the halves meet.  inv_emodulation's last step + shiftrountine_with_rnd (generic:869, :1638) --- */
  /* this wave's channel: 9 slots of history from the ring, in flight across the exchange */
  int16_t h_lo[9], h_hi[9];
#pragma unroll
  for (int A = 1; A <= 9; A++) {
    int pos = d_old + 128 * A;
    if (pos >= RING) pos -= RING;
    h_lo[A - 1] = st_w->ring[pos + lane];
    h_hi[A - 1] = st_w->ring[pos + 64 + lane];
  }
  int32_t o[64];
#pragma unroll
  for (int c = 0; c < 2; c++) { /* the lanes of channel c hand their half over through the (half-size) tiles */
    if (lch == c) {
#pragma unroll
      for (int k = 0; k < 64; k++) tile_own[RS * lrow + k] = x[k];
    }
    __syncthreads();
    if (lch == c) {
#pragma unroll
      for (int k = 0; k < 64; k++) o[k] = tile_oth[RS * lrow + k];
    }
    __syncthreads(); /* after the second round both tiles are dead: the pair rows may overwrite them */
  }
  {
    const int ch = lane >> 5, slot = lane & 31;
    const int shift = -((ch ? st_syn1 : st_syn0) - 3) + 1;
    const int32_t hi = FX_MAX32 >> shift, lo = FX_MIN32 >> shift;
    /* round16(shl_sat(a, b)) == round16(clamp(a, MIN >> b, MAX >> b) << b): the clamped value's top 17 bits decide */
    int16_t *row = reinterpret_cast<int16_t *>(E + (ch * EROWS + 9 + slot) * RS);
    if (w == 0) { /* x = real half, o = imaginary half: b[c] -> low half of E[slot][c] */
#pragma unroll
      for (int c = 0; c < 64; c++) {
        int32_t a = fx_sub_sat(o[c], x[c]);
        a = clamp_med3(a, lo, hi);
        row[2 * c] = fx_round16(fx_shlw(a, shift));
      }
    } else { /* x = imaginary half, o = real half: b[64 + c] -> high half of E[slot + 1][c] */
#pragma unroll
      for (int c = 0; c < 64; c++) {
        int32_t a = fx_add_sat(x[63 - c], o[63 - c]);
        a = clamp_med3(a, lo, hi);
        row[2 * RS + 2 * c + 1] = fx_round16(fx_shlw(a, shift));
      }
    }
    int16_t *hrow = reinterpret_cast<int16_t *>(E + (w * EROWS) * RS);
#pragma unroll
    for (int A = 1; A <= 9; A++) { /* slot -A: row 9 - A low halves, row 10 - A high halves */
      hrow[(9 - A) * 2 * RS + 2 * lane] = h_lo[A - 1];
      hrow[(10 - A) * 2 * RS + 2 * lane + 1] = h_hi[A - 1];
    }
  }
  __syncthreads();
  /* ---- phase C: window-add, wave w = slots 16 w .. 16 w + 15 of both channels, lane = sample k ----------------- */
  {
    xq_short2 cp[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
      cp[j].x = xaac_qmf_qmf_c[128 * j + lane];
      cp[j].y = xaac_qmf_qmf_c[128 * j + 64 + lane];
    }
    int32_t e[2][24]; /* E[16 w - 8 .. 16 w + 15][lane] */
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int q = 0; q < 24; q++) e[c][q] = E[(c * EROWS + 9 + 16 * w - 8 + q) * RS + lane];
    int32_t *out = reinterpret_cast<int32_t *>(p.pcm) + (size_t)i * 2048 + 1024 * w + lane; /* one word per L,R pair */
#pragma unroll
    for (int s = 0; s < 16; s++) {
      int32_t acc[2] = {0x4000, 0x4000};
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int j = 0; j < 5; j++) acc[c] = __builtin_amdgcn_sdot2(__builtin_bit_cast(xq_short2, e[c][8 + s - 2 * j]), cp[j], acc[c], false);
      const uint32_t l = (uint32_t)(uint16_t)(fx_add_sat(acc[0], acc[0]) >> 16), r = (uint32_t)(uint16_t)(fx_add_sat(acc[1], acc[1]) >> 16);
      if (!inactive0 && !inactive1) {
        out[64 * s] = (int32_t)(l | (r << 16));
      } else { /* one channel only: its samples alone (the other's are left as they are) */
        int16_t *o16 = reinterpret_cast<int16_t *>(out + 64 * s);
        if (!inactive0) o16[0] = (int16_t)l;
        if (!inactive1) o16[1] = (int16_t)r;
      }
    }
  }
  /* ---- state of channel w: ring blocks of the last 10 slots, drc offset, window phase --------------------------- */
  if (w == 0 ? inactive0 : inactive1) return;
  {
    const int d_new = (d_old + RING - (32 * 128) % RING) % RING; /* 32 slots of 128 downwards */
    const int ph_new = (st_w->phase + 128) % 640;
    const int16_t *hrow = reinterpret_cast<const int16_t *>(E + (w * EROWS) * RS);
    int16_t v_lo[10], v_hi[10];
#pragma unroll
    for (int A = 1; A <= 10; A++) { /* age relative to the NEXT frame's slot 0: slot 32 - A */
      v_lo[A - 1] = hrow[(9 + 32 - A) * 2 * RS + 2 * lane];
      v_hi[A - 1] = hrow[(10 + 32 - A) * 2 * RS + 2 * lane + 1];
    }
#pragma unroll
    for (int A = 1; A <= 10; A++) {
      int pos = d_new + 128 * A;
      if (pos >= RING) pos -= RING;
      if (pos >= RING) pos -= RING;
      st_w->ring[pos + lane] = v_lo[A - 1];
      st_w->ring[pos + 64 + lane] = v_hi[A - 1];
    }
    if (lane == 0) {
      st_w->drc_offset = (int16_t)d_new;
      st_w->phase = (int16_t)ph_new;
    }
  }
}

extern "C" hipError_t xaac_launch_qmf_synthesis_pair(const XaacQmfSynPairParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_qmf_synthesis_pair_kernel, dim3(p->n), dim3(128), XAAC_QMF_SYN_PAIR_LDS, stream, *p);
  return hipGetLastError();
}

/* ===================================================================================== */
/* The LD / ELD flavour of the complex analysis bank.  qmf_c_eld3 is periodic in the 320 taps of the low-delay prototype and
   the reference's four pointers (ring write position, two filter offsets, the ring half fp1 starts in) rotate together
   through ten phases, so the window-add is again time invariant (derived by simulating the pointer state machine of
   generic:662-741, checked against the literal oracle):
     z[s][m] = sum_{j<5} x[32 s + 31 - (m + 64 j)] * c3[m + 64 j],  m = 0..63
   (sum |c3| over a branch <= 18982: the saturating adds of qmf_dec.c:484-535 cannot clip).  One wave = FOUR channel-frames
   of 16 (or 15) slots: window-add with lanes = the 64 branches, then one lane = one slot for the transform. */
namespace {
__device__ __forceinline__ int eld_phase(const xaac_qmf_ana_eld_state *st) { /* 0..9, or -1 for pointers out of step */
  const int wr = st->wr;
  if (wr < 0 || wr > 288 || (wr & 31)) return -1;
  const int t = ((320 - wr) / 32) % 10;
  const int f1 = (t & 1) ? 32 * (t + 1) : 32 * t, f2 = (t & 1) ? 32 * t : 32 * t + 32;
  return (st->f1 == f1 && st->f2 == f2 && st->fp == 32 * (t & 1)) ? t : -1;
}
}  // namespace

__global__ __launch_bounds__(64) void xaac_qmf_analysis_eld_kernel(xaac_qmf_ana_eld_batch p, XaacQmfEldChain cn) {
  const auto state_of = [&](int ch) {
    return reinterpret_cast<xaac_qmf_ana_eld_state *>(reinterpret_cast<char *>(p.state) +
                                                      (size_t)ch * (cn.state_stride ? cn.state_stride : (int)sizeof(xaac_qmf_ana_eld_state)));
  };
  const int fac = cn.pcm_ch_fac > 1 ? cn.pcm_ch_fac : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int H = 288 + 512;
  int16_t *hist = reinterpret_cast<int16_t *>(smem);              /* [4][H], oldest first */
  int32_t *z = reinterpret_cast<int32_t *>(smem + 4 * H * 2);     /* [64][65] */
  const int lane = threadIdx.x, ns = p.n_slots, quad = blockIdx.x;
  int32_t coef[5];
#pragma unroll
  for (int j = 0; j < 5; j++) coef[j] = xaac_qmf_eld_c3[lane + 64 * j];
  int phase[4];
  for (int c = 0; c < 4; c++) {
    const int ch = 4 * quad + c;
    int16_t *h = hist + c * H;
    phase[c] = -1;
    if (ch < p.n_ch) phase[c] = eld_phase(state_of(ch));
    if (phase[c] >= 0) {
      const xaac_qmf_ana_eld_state *st = state_of(ch);
      const int wr = st->wr;
      for (int a = lane; a < 288; a += 64) h[287 - a] = st->ring[ana_ring_pos(wr, a)];
      const int16_t *src = p.pcm + (size_t)(ch / fac) * 32 * ns * fac + ch % fac;
      for (int i = lane; i < 32 * ns; i += 64) h[288 + i] = src[(size_t)i * fac];
    } else {
      for (int i = lane; i < H; i += 64) h[i] = 0;
    }
    if (lane == 0 && p.status && ch < p.n_ch) p.status[ch] = phase[c] >= 0 ? 0 : -1;
  }
  __syncthreads();
  for (int r = 0; r < 64; r++) { /* row r = channel r / 16, slot r % 16 */
    const int s = r & 15;
    int32_t acc = 0;
    if (s < ns) {
      const int16_t *h = hist + (r >> 4) * H + 288 + 32 * s + 31 - lane;
#pragma unroll
      for (int j = 0; j < 5; j++) acc += (int32_t)h[-64 * j] * coef[j];
    }
    z[65 * r + lane] = acc;
  }
  __syncthreads();
  {
    int32_t in[64], sb[128], t[128];
#pragma unroll
    for (int k = 0; k < 64; k++) in[k] = z[65 * lane + k];
    int usb = p.usb;
    if (cn.frame) { /* the lane's channel: what ixheaacd_rescale_x_overlap leaves in the bank (sbrdec_lpfuncs.c:470) */
      const int chl = 4 * quad + (lane >> 4);
      if (chl < p.n_ch) {
        const xaac_sbr_frame *fr = cn.frame + chl;
        usb = fr->apply_processing ? fr->max_qmf_subband_aac
                                   : *reinterpret_cast<const int16_t *>(reinterpret_cast<const char *>(cn.codec_usb) + (size_t)chl * cn.state_stride);
        usb = usb < 0 ? 0 : (usb > 32 ? 32 : usb);
      }
    }
    xq_fwd_modulation(in, sb, t, usb, true);
#pragma unroll
    for (int k = 0; k < 32; k++) {
      z[65 * lane + k] = sb[k];
      z[65 * lane + 32 + k] = sb[64 + k];
    }
  }
  __syncthreads();
  for (int r = 0; r < 64; r++) {
    const int c = r >> 4, s = r & 15, ch = 4 * quad + c;
    if (ch >= p.n_ch || phase[c] < 0 || s >= ns) continue;
    int32_t *row = p.qmf + ((size_t)ch * ns + s) * p.slot_stride;
    row[(lane & 31) + 64 * (lane >> 5)] = z[65 * r + lane];
  }
  for (int c = 0; c < 4; c++) { /* the ring and the pointers as the reference leaves them after n_slots slots */
    const int ch = 4 * quad + c;
    if (ch >= p.n_ch || phase[c] < 0) continue;
    xaac_qmf_ana_eld_state *st = state_of(ch);
    const int t = (phase[c] + ns) % 10, wr_new = (320 - 32 * t) % 320;
    const int16_t *h = hist + c * H + 288 + 32 * ns; /* one past the newest sample */
    for (int a = lane; a < 320; a += 64) st->ring[ana_ring_pos(wr_new, a)] = h[-1 - a];
    if (lane == 0) {
      st->wr = (int16_t)wr_new;
      st->f1 = (int16_t)((t & 1) ? 32 * (t + 1) : 32 * t);
      st->f2 = (int16_t)((t & 1) ? 32 * t : 32 * t + 32);
      st->fp = (int16_t)(32 * (t & 1));
    }
  }
}

/* The LD / ELD synthesis bank.  As for the AAC bank the pointer rotation nets out (simulated; the oracle is the literal
   form): y[s][k] = rnd + sum_{A<10} v[s-A][64 (A&1) + k] * c[64 A + k] on qmf_c_eld (sum |c| <= 27506: no clipping before
   the final saturating shift), v[s] = the 128 ring samples slot s writes.  One wave = four channel-frames: lane =
   (channel, slot) for the slot transforms, then lanes = the 64 outputs of a row.  Ten phases of the four state words. */
namespace {
__device__ __forceinline__ int eld_syn_phase(const xaac_qmf_syn_eld_state *st) {
  const int d = st->drc_offset;
  if (d < 0 || d > 1152 || (d & 127)) return -1;
  const int t = ((1280 - d) / 128) % 10;
  return (st->phase == 64 * t && st->fp == 64 * (t & 1) && st->sixty4 == ((t & 1) ? -64 : 64)) ? t : -1;
}
}  // namespace

__global__ __launch_bounds__(64) void xaac_qmf_synthesis_eld_kernel(xaac_qmf_syn_eld_batch p, XaacQmfEldChain cn) {
  const auto state_of = [&](int ch) {
    return reinterpret_cast<xaac_qmf_syn_eld_state *>(reinterpret_cast<char *>(p.state) +
                                                      (size_t)ch * (cn.state_stride ? cn.state_stride : (int)sizeof(xaac_qmf_syn_eld_state)));
  };
  const int fac = cn.pcm_ch_fac > 1 ? cn.pcm_ch_fac : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int VR = 130, VS = 25; /* padded row (int16), rows per channel: 9 of history + up to 16 slots */
  int16_t *v = reinterpret_cast<int16_t *>(smem); /* [4][VS][VR] */
  const int lane = threadIdx.x, ns = p.n_slots, quad = blockIdx.x;
  int32_t coef[10];
#pragma unroll
  for (int a = 0; a < 10; a++) coef[a] = xaac_qmf_eld_c[64 * a + lane];
  int phase[4];
  for (int c = 0; c < 4; c++) {
    const int ch = 4 * quad + c;
    phase[c] = ch < p.n_ch ? eld_syn_phase(state_of(ch)) : -1;
    const bool skipped = ch < p.n_ch && cn.syn_par && cn.syn_par[8 * (size_t)ch + 6] != 0; /* the chain's core refused the frame */
    if (lane == 0 && p.status && ch < p.n_ch && !skipped && (!cn.syn_par || phase[c] < 0)) p.status[ch] = phase[c] >= 0 ? 0 : -1;
    if (skipped) phase[c] = -1;
    int16_t *vc = v + c * VS * VR;
    if (phase[c] >= 0) {
      const xaac_qmf_syn_eld_state *st = state_of(ch);
      const int d0 = st->drc_offset;
      for (int e = lane; e < 9 * 128; e += 64) { /* slot -A (A = 1..9) lives 128 A behind the write offset */
        const int a = 1 + e / 128, o = e % 128;
        vc[(9 - a) * VR + o] = st->ring[(d0 + 128 * a + o) % 1280];
      }
    } else {
      for (int e = lane; e < 9 * VR; e += 64) vc[e] = 0;
    }
  }
  {
    const int c = lane >> 4, s = lane & 15, ch = 4 * quad + c;
    if (phase[c] >= 0 && s < ns) {
      const int16_t *sf = cn.syn_par ? cn.syn_par + 8 * (size_t)ch : p.scale + 4 * (size_t)ch;
      const int lsb = cn.syn_par ? sf[4] : p.lsb, usb = cn.syn_par ? sf[5] : p.usb;
      const int st_syn = sf[3];
      const int ov_lb_shift = (st_syn - sf[1]) - 7, lb_shift = (st_syn - sf[0]) - 7, hb_shift = (st_syn - sf[2]) - 7;
      const int32_t *row = p.qmf + ((size_t)ch * ns + s) * p.slot_stride;
      int32_t x[128], t[128];
      int16_t b[128];
#pragma unroll
      for (int k = 0; k < 128; k++) {
        const int band = k & 63;
        int32_t val = row[k];
        if (band < lsb) val = adj_scale(val, s < p.split ? ov_lb_shift : lb_shift);
        else if (band < usb) val = adj_scale(val, hb_shift);
        x[k] = val;
      }
      if (p.qmf_scaled) {
        int32_t *srow = p.qmf_scaled + ((size_t)ch * ns + s) * p.slot_stride;
#pragma unroll
        for (int k = 0; k < 128; k++) srow[k] = x[k];
      }
      xq_synth_eld_slot(x, t, b, -(st_syn - 3));
      int16_t *dst = v + (c * VS + 9 + s) * VR;
#pragma unroll
      for (int k = 0; k < 128; k++) dst[k] = b[k];
    }
  }
  __syncthreads();
  for (int r = 0; r < 64; r++) {
    const int c = r >> 4, s = r & 15, ch = 4 * quad + c;
    if (phase[c] < 0 || s >= ns) continue;
    const int16_t *vc = v + c * VS * VR;
    int32_t acc = 0x8000 >> 2;
#pragma unroll
    for (int a = 0; a < 10; a++) acc += (int32_t)vc[(9 + s - a) * VR + 64 * (a & 1) + lane] * coef[a];
    p.pcm[(size_t)(ch / fac) * 64 * ns * fac + ((size_t)s * 64 + lane) * fac + ch % fac] = (int16_t)(fx_shl_sat(acc, 2) >> 16);
  }
  for (int c = 0; c < 4; c++) { /* the ring = the last ten slots' samples at the offsets the reference wrote them to */
    const int ch = 4 * quad + c;
    if (phase[c] < 0) continue;
    xaac_qmf_syn_eld_state *st = state_of(ch);
    const int d0 = st->drc_offset, t = (phase[c] + ns) % 10;
    const int16_t *vc = v + c * VS * VR;
    for (int e = lane; e < 1280; e += 64) {
      const int a = e / 128, o = e % 128, s = ns - 1 - a; /* slot s was written at d0 - 128 s */
      st->ring[((d0 - 128 * s) % 1280 + 1280) % 1280 + o] = vc[(9 + s) * VR + o];
    }
    if (lane == 0) {
      st->drc_offset = (int16_t)((1280 - 128 * t) % 1280);
      st->phase = (int16_t)(64 * t);
      st->fp = (int16_t)(64 * (t & 1));
      st->sixty4 = (int16_t)((t & 1) ? -64 : 64);
    }
  }
}

extern "C" hipError_t xaac_launch_qmf_synthesis_eld_chain(const xaac_qmf_syn_eld_batch *p, const XaacQmfEldChain *c, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_qmf_synthesis_eld_kernel, dim3((p->n_ch + 3) / 4), dim3(64), XAAC_QMF_ELD_SYN_LDS, stream, *p, *c);
  return hipGetLastError();
}
extern "C" hipError_t xaac_launch_qmf_synthesis_eld(const xaac_qmf_syn_eld_batch *p, hipStream_t stream) {
  const XaacQmfEldChain none = {};
  return xaac_launch_qmf_synthesis_eld_chain(p, &none, stream);
}

extern "C" hipError_t xaac_launch_qmf_analysis_eld_chain(const xaac_qmf_ana_eld_batch *p, const XaacQmfEldChain *c, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_qmf_analysis_eld_kernel, dim3((p->n_ch + 3) / 4), dim3(64), XAAC_QMF_ELD_LDS, stream, *p, *c);
  return hipGetLastError();
}
extern "C" hipError_t xaac_launch_qmf_analysis_eld(const xaac_qmf_ana_eld_batch *p, hipStream_t stream) {
  const XaacQmfEldChain none = {};
  return xaac_launch_qmf_analysis_eld_chain(p, &none, stream);
}

extern "C" hipError_t xaac_launch_qmf_analysis(const XaacQmfAnaParams *p, int grid, hipStream_t stream) {
  if (p->low_pow)
    hipLaunchKernelGGL(xaac_qmf_analysis_kernel<true>, dim3(grid), dim3(XAAC_QMF_BLOCK),
                       XAAC_QMF_WAVES * XAAC_QMF_ANA_LDS_PER_WAVE, stream, *p);
  else /* two channels per workgroup, not persistent: `grid` (sized for the low-power kernel) does not apply */
    hipLaunchKernelGGL(xaac_qmf_analysis_hq_kernel, dim3((p->n_ch + 1) / 2), dim3(128), 0, stream, *p);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_qmf_synthesis(const XaacQmfSynParams *p, int grid, hipStream_t stream) {
  if (p->low_pow && p->down_sample)
    hipLaunchKernelGGL((xaac_qmf_synthesis_kernel<true, true>), dim3(grid), dim3(XAAC_QMF_BLOCK),
                       XAAC_QMF_WAVES * XAAC_QMF_SYN_LDS_PER_WAVE_LP, stream, *p);
  else if (p->low_pow)
    hipLaunchKernelGGL((xaac_qmf_synthesis_kernel<true, false>), dim3(grid), dim3(XAAC_QMF_BLOCK),
                       XAAC_QMF_WAVES * XAAC_QMF_SYN_LDS_PER_WAVE_LP, stream, *p);
  else if (p->down_sample)
    hipLaunchKernelGGL((xaac_qmf_synthesis_kernel<false, true>), dim3(grid), dim3(XAAC_QMF_BLOCK),
                       XAAC_QMF_WAVES * XAAC_QMF_SYN_LDS_PER_WAVE_HQ, stream, *p);
  else
    hipLaunchKernelGGL((xaac_qmf_synthesis_kernel<false, false>), dim3(grid), dim3(XAAC_QMF_BLOCK),
                       XAAC_QMF_WAVES * XAAC_QMF_SYN_LDS_PER_WAVE_HQ, stream, *p);
  return hipGetLastError();
}

extern "C" int xaac_qmf_blocks_per_cu(int which) {
  int n = 0;
  hipError_t e;
  switch (which) {
    case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, xaac_qmf_analysis_kernel<true>, XAAC_QMF_BLOCK, XAAC_QMF_WAVES * XAAC_QMF_ANA_LDS_PER_WAVE); break;
    case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, xaac_qmf_analysis_kernel<false>, XAAC_QMF_BLOCK, XAAC_QMF_WAVES * XAAC_QMF_ANA_LDS_PER_WAVE); break;
    case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (xaac_qmf_synthesis_kernel<true, false>), XAAC_QMF_BLOCK, XAAC_QMF_WAVES * XAAC_QMF_SYN_LDS_PER_WAVE_LP); break;
    default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (xaac_qmf_synthesis_kernel<false, false>), XAAC_QMF_BLOCK, XAAC_QMF_WAVES * XAAC_QMF_SYN_LDS_PER_WAVE_HQ); break;
  }
  if (e != hipSuccess || n < 1) n = 1;
  return n;
}

/* xaac_warm_up (xaac_abi.cpp): asking for a kernel's attributes puts this translation unit's code object on the device */
extern "C" hipError_t xaac_warm_sbr_qmf(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&xaac_qmf_synthesis_pair_kernel));
}
