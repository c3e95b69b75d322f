/*
 * esbr_qmf_kernel.hip -- gfx950 kernels for the QMF banks of the reference's default SBR path ("Path A", -esbr:1):
 *   xaac_esbr_analysis_kernel   <->  ixheaacd_esbr_analysis_filt_block   decoder/ixheaacd_sbr_dec.c:185 (32 channels)
 *   xaac_esbr_synthesis_kernel  <->  the bank loop of ixheaacd_esbr_synthesis_filt_block   sbr_dec.c:572-656 (64 channels)
 * These banks are float at their edges and integer inside: core samples x 2^15 -> WORD32, a 32-bit prototype filter with
 * 64-bit accumulation, the transforms of sbr_qmf.h on 32-bit twiddles with 64-bit products, results x 2^-8 / x 2^-16 back
 * to float.  Every conversion is exact, so the outputs are bit-identical to the reference's, like the fixed-point banks'.
 *
 * Mapping: the same as sbr_qmf_kernel.hip -- the banks are time-invariant polyphase FIRs around a per-slot transform, so
 * every slot is independent given the frame's samples and the history the ring holds (the closed forms there carry over:
 * the ring / window-phase state machines are the same, only the word size differs).  One wave = two channel-frames:
 * window-add with lanes = polyphase outputs, transform with lane = slot, rows through a padded LDS tile, the WORD32 ring
 * kept word-identical with the reference's as the persistent state.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sbr_qmf.h"
#include "esbr_qmf_kernel.h"

namespace {

constexpr int kHist = 288 + 1024;

__device__ __forceinline__ int ring_pos(int wr, int a) { /* where the ring keeps the sample of age a (0 = newest) */
  int p = wr + 32 + a;
  return p >= 320 ? p - 320 : p;
}

/* 32 steps of the window-pointer bookkeeping of sbr_dec.c:262-277 (it does not influence the samples) */
__device__ __forceinline__ int win_after_frame(int w) {
  int f1 = w, f2 = w + 64;
  for (int s = 0; s < 32; s++) {
    f1 += 64;
    f2 += 64;
    const int t = f1;
    f1 = f2;
    f2 = t;
    if (f2 > 640) {
      f1 = 0;
      f2 = 64;
    }
  }
  return f1;
}

}  // namespace

__global__ __launch_bounds__(64) void xaac_esbr_analysis_kernel(XaacEsbrAnaParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  int32_t *hist = reinterpret_cast<int32_t *>(smem);                 /* [2][kHist] time-ordered, oldest first */
  int32_t *z = reinterpret_cast<int32_t *>(smem);                    /* [64][65]: the exchange tile takes the history's place once
                                                                        the window-add has read it and the rings are written */
  const int pair = blockIdx.x;
  int32_t coef[5]; /* c[2 m + 128 j] of this lane's polyphase branch m = lane */
#pragma unroll
  for (int j = 0; j < 5; j++) coef[j] = xaac_qmf_esbr_qmf_c[2 * lane + 128 * j];
  /* Everything the pair reads is in flight together: both channels' ring positions and window offsets, their rings (fetched by
     position, not by age: the loads then do not wait for the position, which only says where in the time-ordered history a
     fetched sample belongs) and their 1024 core samples.  (One after the other -- position, then ring, then samples, channel by
     channel -- the workgroup paid half a dozen memory round trips before its first multiply.) */
  int pos_v[2] = {0, 0}, win_v[2] = {0, 0};
  int32_t rg[2][5];
  float sv[2][16];
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const int ch = 2 * pair + c;
    if (ch < p.n_ch) { /* (uniform) */
      const xaac_esbr_ana_state *st = reinterpret_cast<const xaac_esbr_ana_state *>(reinterpret_cast<const char *>(p.state) + (size_t)ch * p.state_stride);
      const float *src = p.core + (size_t)ch * 1024;
      pos_v[c] = st->pos;
      win_v[c] = st->win_off;
#pragma unroll
      for (int j = 0; j < 5; j++) rg[c][j] = st->ring[lane + 64 * j];
#pragma unroll
      for (int j = 0; j < 16; j++) sv[c][j] = src[lane + 64 * j];
    }
  }
  int wr_c[2];
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const int ch = 2 * pair + c;
    int32_t *h = hist + c * kHist;
    int wr = __builtin_amdgcn_readfirstlane(pos_v[c]);
    wr = ((wr % 320 + 320) % 320) & ~31; /* a block start (the position moves by 32 from 0) */
    wr_c[c] = wr;
    if (ch < p.n_ch) {
#pragma unroll
      for (int j = 0; j < 5; j++) {
        int a = lane + 64 * j - wr - 32; /* age of the sample at this position: ring_pos(wr, a) = position */
        a += a < 0 ? 320 : 0;
        a += a < 0 ? 320 : 0;
        if (a < 288) h[287 - a] = rg[c][j];
      }
#pragma unroll
      for (int j = 0; j < 16; j++) h[288 + lane + 64 * j] = fx_f2i_trunc(sv[c][j] * 32768.0f); /* sbr_dec.c:248 */
    } else {
      for (int i = lane; i < kHist; i += 64) h[i] = 0;
    }
  }
  __syncthreads();
  /* window-add (ixheaacd_esbr_qmfanal32_winadd, qmf_dec.c:537): 64-bit sums, >> 31 */
  int32_t wa[64]; /* this lane's polyphase branch of all 64 slots */
#pragma unroll
  for (int r = 0; r < 64; r++) {
    const int32_t *h = hist + (r >> 5) * kHist + 288 + 32 * (r & 31) + 31 - lane;
    int64_t acc = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) acc = xq_add64(acc, (int64_t)h[-64 * j] * coef[j]);
    wa[r] = (int32_t)(acc >> 31);
  }
  for (int c = 0; c < 2; c++) { /* state: the ring as the reference leaves it after 32 slots (the history's last 320 samples) */
    const int ch = 2 * pair + c;
    if (ch >= p.n_ch) break;
    xaac_esbr_ana_state *st = reinterpret_cast<xaac_esbr_ana_state *>(reinterpret_cast<char *>(p.state) + (size_t)ch * p.state_stride);
    const int wr_new = (wr_c[c] + 256) % 320;
    const int w_new = win_after_frame(__builtin_amdgcn_readfirstlane(win_v[c]));
    const int32_t *h = hist + c * kHist;
    int32_t keep[5];
#pragma unroll
    for (int j = 0; j < 5; j++) keep[j] = h[kHist - 1 - (lane + 64 * j)];
    __syncthreads(); /* every lane has read the old ring positions in the copy-in above */
#pragma unroll
    for (int j = 0; j < 5; j++) st->ring[ring_pos(wr_new, lane + 64 * j)] = keep[j];
    if (lane == 0) {
      st->pos = wr_new;
      st->win_off = w_new;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 64; r++) z[65 * r + lane] = wa[r];
  __syncthreads();
  { /* per-slot transform, lane = slot */
    int32_t in[64], sb[128], t[128];
#pragma unroll
    for (int k = 0; k < 64; k++) in[k] = z[65 * lane + k];
    xq_esbr_fwd_modulation(in, sb, t);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k++) {
      z[65 * lane + k] = sb[k];
      z[65 * lane + 32 + k] = sb[64 + k];
    }
  }
  __syncthreads();
  for (int r = 0; r < 64; r++) { /* rows out: 32 real and 32 imaginary bands per slot, x 1/256 (sbr_dec.c:285-290) */
    const int ch = 2 * pair + (r >> 5);
    if (ch >= p.n_ch) break;
    const float v = (float)z[65 * r + lane] * (1.0f / 256.0f);
    float *row = (lane < 32 ? p.qmf_re : p.qmf_im) + ((size_t)ch * 32 + (r & 31)) * 64;
    row[lane & 31] = v;
  }
}

/* The 24- and 16-channel banks of 8:3 and 4:1 SBR (sbr_dec.c:213-236): lane = time slot -- 64 slots of 16 samples fill a wave,
   32 slots of 24 samples half of one, so a wave takes two 24-channel channel-frames.  Nothing here is closed-form: a lane forms
   its slot exactly as the reference's loop does at that slot -- the ring as it stands then (per block of NB words: the newest
   write of this frame to it, or the words the state holds), the window pointers after that many steps -- so any state the
   reference can hand over gives the reference's result.  The ring is ten blocks, a slot's window-add reads five of them per
   half: where a block's words come from is worked out once per block, not per tap. */
namespace {
template <int NB>
struct XeRingAt {        /* anal_filter_states_32 at slot s of this frame */
  const int32_t *old_ring; /* LDS: the state's ring */
  const int32_t *frame;    /* LDS: (WORD32)(core * 2^15), slot k's NB samples at k (NB + 1): lanes are slots, and rows NB words
                              apart would put all of a wave's reads of one tap on two (NB 16) or four (NB 24) of the 32 banks */
  int pb, s;               /* block the frame's first slot writes; the slot */
  /* block b at slot s: its word r is base[dir * r] */
  __device__ __forceinline__ void block(int b, const int32_t *&base, int &dir) const {
    int k0 = pb - b;
    k0 += k0 < 0 ? 10 : 0;                              /* first slot of the frame that writes block b; then every tenth */
    if (k0 > s) {
      base = old_ring + b * NB;
      dir = 1;
    } else {
      const int k = k0 + 10 * ((s - k0) / 10);
      base = frame + k * (NB + 1) + NB - 1;             /* sbr_dec.c:247-250: the block holds the slot's samples reversed */
      dir = -1;
    }
  }
  __device__ __forceinline__ int32_t operator()(int pos) const {
    const int b = pos / NB;
    const int32_t *base;
    int dir;
    block(b, base, dir);
    return base[dir * (pos - b * NB)];
  }
};
}  // namespace

template <int NB>
__global__ __launch_bounds__(64) void xaac_esbr_analysis_nb_kernel(XaacEsbrAnaNbParams p) {
  constexpr int RS = 2 * NB + 1, CPW = NB == 24 ? 2 : 1, SL = 64 / CPW; /* tile row: NB real | NB imaginary words (+ 1: no bank conflicts); channel-frames a wave takes, lanes (slots) of each */
  __shared__ int32_t old_ring[CPW][10 * NB];
  __shared__ int32_t frame[CPW][(1024 / NB) * (NB + 1)];
  __shared__ int32_t tile[64 * RS];
  /* the window: a lane's offsets into it depend on its slot.  With 64 slots a wave the reads go through LDS (169 -> 98 us per 8192
     channel-frames); with two channel-frames of 32 slots the vector cache serves them as well and the LDS is better spent on
     occupancy (87 us against 94) */
  constexpr int NWIN = NB == 24 ? 1 : 1280;
  __shared__ int32_t win_lds[NWIN];
  const int lane = threadIdx.x, n_slots = p.n_slots;
  if (NB != 24)
    for (int i = lane; i < NWIN; i += 64) win_lds[i] = XqEsbrAna<NB>::win()[i];
  const int lc = lane / SL, slot = lane % SL, my_ch = (int)blockIdx.x * CPW + lc;
  constexpr int fo = XqEsbrAna<NB>::fo;
  int pb_c[CPW], win_c[CPW];
#pragma unroll
  for (int c = 0; c < CPW; c++) {
    const int ch = (int)blockIdx.x * CPW + c;
    pb_c[c] = win_c[c] = 0;
    if (ch >= p.n_ch) continue; /* (uniform) */
    const xaac_esbr_ana_state *st = reinterpret_cast<const xaac_esbr_ana_state *>(reinterpret_cast<const char *>(p.state) + (size_t)ch * p.state_stride);
    const float *src = p.core + (size_t)ch * p.core_stride;
    int pos0 = st->pos, win0 = st->win_off;
    for (int i = lane; i < 10 * NB; i += 64) old_ring[c][i] = st->ring[i];
    for (int i = lane; i < NB * n_slots; i += 64) frame[c][i + i / NB] = fx_f2i_trunc(src[i] * 32768.0f); /* sbr_dec.c:248 */
    pos0 = __builtin_amdgcn_readfirstlane(pos0);
    win0 = __builtin_amdgcn_readfirstlane(win0);
    /* states no run of the reference produces (a position off the block grid, a window offset off its step) are brought onto
       the grid instead of being followed out of the arrays */
    int pb = pos0 / NB;
    pb_c[c] = pb < 0 ? 0 : (pb > 9 ? 9 : pb);
    win_c[c] = win0 < 0 ? 0 : (win0 > 9 * fo ? 9 * fo : win0 / fo * fo);
  }
  __syncthreads();
  const int pb = CPW == 2 && lc ? pb_c[CPW - 1] : pb_c[0], win0 = CPW == 2 && lc ? win_c[CPW - 1] : win_c[0];
  if (slot < n_slots && my_ch < p.n_ch) {
    int w1 = win0, w2 = win0 + fo;
    for (int k = 0; k < slot; k++) xq_esbr_win_step<NB>(w1, w2);
    const XeRingAt<NB> rg = {old_ring[lc], frame[lc], pb, slot};
    const int32_t *bp[10];
    int bd[10];
#pragma unroll
    for (int b = 0; b < 10; b++) rg.block(b, bp[b], bd[b]);
    /* ixheaacd_esbr_qmfanal32_winadd (qmf_dec.c:537): the first NB outputs from the ring at offset f1 with the window at w1, the
       second NB at f2 / w2; f1 is 0 in even slots and NB in odd ones (sbr_dec.c:262-265): blocks 2 j + odd | 2 j + 1 - odd */
    const int odd = slot & 1;
    const int32_t *cw = NB == 24 ? XqEsbrAna<NB>::win() : win_lds;
    constexpr int cs = XqEsbrAna<NB>::cs;
    int32_t anal[2 * NB], sb[128], t[128];
#pragma unroll
    for (int n = 0; n < NB; n++) {
      int64_t a1 = 0, a2 = 0;
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const int32_t *p1 = odd ? bp[2 * j + 1] : bp[2 * j], *p2 = odd ? bp[2 * j] : bp[2 * j + 1];
        const int d1 = odd ? bd[2 * j + 1] : bd[2 * j], d2 = odd ? bd[2 * j] : bd[2 * j + 1];
        a1 = xq_add64(a1, (int64_t)p1[d1 * n] * cw[w1 + cs * (n + 2 * NB * j)]);
        a2 = xq_add64(a2, (int64_t)p2[d2 * n] * cw[w2 + cs * (n + 2 * NB * j)]);
      }
      anal[n] = (int32_t)(a1 >> 31);
      anal[NB + n] = (int32_t)(a2 >> 31);
    }
    xq_esbr_fwd_modulation_nb<NB>(anal, sb, t);
#pragma unroll
    for (int k = 0; k < NB; k++) {
      tile[RS * lane + k] = sb[k];
      tile[RS * lane + NB + k] = sb[64 + k];
    }
  }
  __syncthreads();
  { /* rows out: lanes 0..31 the real bands, 32..63 the imaginary ones; bands NB..31 are never written by the reference and are
       zero in its buffers: written as zeros here, for the kernels behind that read 32 bands of a row */
    const float gain = XqEsbrAna<NB>::gain();
    const int k = lane & 31;
#pragma unroll
    for (int c = 0; c < CPW; c++) {
      const int ch = (int)blockIdx.x * CPW + c;
      if (ch >= p.n_ch) continue;
      float *dst = (lane < 32 ? p.qmf_re : p.qmf_im) + (size_t)ch * p.out_stride;
      for (int r = 0; r < n_slots; r++) dst[64 * r + k] = k < NB ? (float)tile[RS * (c * SL + r) + (lane < 32 ? k : NB + k)] * gain : 0.0f;
    }
  }
#pragma unroll
  for (int c = 0; c < CPW; c++) { /* the state as the reference leaves it: every block's newest write, the pointers after n_slots steps */
    const int ch = (int)blockIdx.x * CPW + c;
    if (ch >= p.n_ch) continue;
    xaac_esbr_ana_state *st = reinterpret_cast<xaac_esbr_ana_state *>(reinterpret_cast<char *>(p.state) + (size_t)ch * p.state_stride);
    const XeRingAt<NB> rg = {old_ring[c], frame[c], pb_c[c], n_slots - 1};
    for (int i = lane; i < 10 * NB; i += 64) st->ring[i] = n_slots > 0 ? rg(i) : old_ring[c][i];
    if (lane == 0) {
      int w1 = win_c[c], w2 = win_c[c] + fo;
      for (int k = 0; k < n_slots; k++) xq_esbr_win_step<NB>(w1, w2);
      int pn = (pb_c[c] - n_slots) % 10;
      pn += pn < 0 ? 10 : 0;
      st->pos = pn * NB;
      st->win_off = w1;
    }
  }
}

/* Two waves per channel pair, as in xaac_qmf_synthesis_pair_kernel (sbr_qmf_kernel.hip): the slot transform's two independent
   halves (sbr_qmf.h: xq_cos_sin_mod_half) run on the workgroup's two waves -- wave h takes the real (h = 0) or imaginary
   (h = 1) half rows of all 64 rows (2 channels x 32 slots, lane = (channel, slot)) through its own half-size tile, a lane
   holds 2 x 64 words instead of three 128-word arrays (243 VGPRs and two waves per SIMD in the one-wave version) -- and
   the halves meet through the tiles: wave 0 forms the ring samples b[0..63] of every slot, wave 1 b[64..127].  The
   window-add of a channel is split by slots (wave h: slots 16 h .. 16 h + 15, lane = sample), history and state by words. */
#ifndef XE_SYN_MIN_WAVES
#define XE_SYN_MIN_WAVES 2
#endif
__global__ __launch_bounds__(128, XE_SYN_MIN_WAVES) void xaac_esbr_synthesis_kernel(XaacEsbrSynParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = 65, VSLOTS = 41, VROW = 129, RING = 1280;
  static_assert(2 * 32 * RS <= VSLOTS * VROW, "the tiles fit where the ring samples go");
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int32_t *tile_own = reinterpret_cast<int32_t *>(smem) + w * 32 * RS;
  const int32_t *tile_oth = reinterpret_cast<const int32_t *>(smem) + (1 - w) * 32 * RS;
  int32_t *v = reinterpret_cast<int32_t *>(smem); /* [VSLOTS][VROW] ring samples of one channel at a time, once the tiles are dead */
  const int pair = blockIdx.x, lch = lane >> 5, lrow = lane & 31;
  int32_t coef[10]; /* c[64 A + k], k = lane */
#pragma unroll
  for (int a = 0; a < 10; a++) coef[a] = xaac_qmf_esbr_qmf_c[64 * a + lane];
  int32_t x[64];
  { /* half rows in (lane = band), (WORD32)(x * 64) (sbr_dec.c:592-595), through the tile to lane = (channel, slot) */
    const float *src = w ? p.qmf_im : p.qmf_re;
    float tmp[64]; /* all 64 half rows in flight: one memory latency */
#pragma unroll
    for (int r = 0; r < 64; r++) {
      const int ch = 2 * pair + (r >> 5);
      tmp[r] = src[(size_t)(ch < p.n_ch ? ch : 0) * p.in_stride + (size_t)(r & 31) * 64 + lane];
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const bool live = 2 * pair + c < p.n_ch;
#pragma unroll
      for (int j = 0; j < 32; j++) tile_own[RS * j + lane] = live ? fx_f2i_trunc(tmp[32 * c + j] * 64.0f) : 0;
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
      xq_cos_sin_mod_half<32, 0, XqW32>(x, t);
    else
      xq_cos_sin_mod_half<32, 1, XqW32>(x, t);
  }
  {
    int32_t o[64];
#pragma unroll
    for (int c = 0; c < 2; c++) { /* the lanes of channel c hand their half over through the tiles */
      if (lch == c) {
#pragma unroll
        for (int k = 0; k < 64; k++) tile_own[RS * lrow + k] = x[k];
      }
      __syncthreads();
      if (lch == c) {
#pragma unroll
        for (int k = 0; k < 64; k++) o[k] = tile_oth[RS * lrow + k];
      }
      __syncthreads(); /* after the second round both tiles are dead: the ring samples may overwrite them */
    }
    /* ixheaacd_esbr_inv_modulation's last step + ixheaacd_shiftrountine_with_rnd_hq (qmf_dec.c:733, generic:1704), shift =
       out_scalefactor + 1 = 6 (sbr_dec.c:556 / :604): this wave's 64 of the slot's 128 ring samples, in place in x */
    if (w == 0) { /* x = real half s[c], o = imaginary half s[64 + c]: b[c] */
#pragma unroll
      for (int c = 0; c < 64; c++) x[c] = fx_shl_sat(fx_sub_sat(o[c], x[c]), 6);
    } else { /* x = imaginary half, o = real half: b[64 + c] = s[64 + 63 - c] + s[63 - c] */
#pragma unroll
      for (int c = 0; c < 32; c++) {
        const int32_t lo = fx_shl_sat(fx_add_sat(x[63 - c], o[63 - c]), 6), hi = fx_shl_sat(fx_add_sat(x[c], o[c]), 6);
        x[c] = lo;
        x[63 - c] = hi;
      }
    }
  }
  for (int c = 0; c < 2; c++) { /* one channel's ring samples in LDS at a time */
    const int ch = 2 * pair + c;
    if (ch >= p.n_ch) break; /* (uniform) */
    /* the right bank of a PS batch: a stream without parametric stereo (channel_mode != PS_STEREO) has no right channel -- no
       output, no state change, as the reference leaves that bank alone until PS starts (uniform over the workgroup) */
    if (p.only_ps && __builtin_amdgcn_readfirstlane((int)p.only_ps[ch].channel_mode) != 3) continue;
    xaac_esbr_syn_state *st = reinterpret_cast<xaac_esbr_syn_state *>(reinterpret_cast<char *>(p.state) + (size_t)ch * p.state_stride);
    if (lch == c) {
      int32_t *dst = v + (9 + lrow) * VROW + 64 * w;
#pragma unroll
      for (int k = 0; k < 64; k++) dst[k] = x[k];
    }
    int d = st->drc_offset;
    d = ((d % RING + RING) % RING) & ~127;
    const int f_old = st->filt_off;
    { /* 9 slots of history from the ring: 1152 words over 128 threads, nine loads in flight */
      int32_t tv[9];
#pragma unroll
      for (int j = 0; j < 9; j++) {
        const int i = (int)threadIdx.x + 128 * j; /* = 128 j + word */
        int pos = d + 128 * (9 - j) + (int)threadIdx.x;
        if (pos >= RING) pos -= RING;
        tv[j] = st->ring[pos];
        (void)i;
      }
#pragma unroll
      for (int j = 0; j < 9; j++) v[j * VROW + (int)threadIdx.x] = tv[j];
    }
    __syncthreads();
    { /* window-add (ixheaacd_esbr_qmfsyn64_winadd, generic:1544), x 2^-16 to float: wave w takes slots 16 w .. 16 w + 15 */
      float *dst = p.out + (size_t)ch * (p.out_stride ? p.out_stride : 2048);
#pragma unroll 4
      for (int s = 16 * w; s < 16 * w + 16; s++) {
        const int32_t *vs = v + (9 + s) * VROW + lane;
        int64_t acc = 0;
#pragma unroll
        for (int A = 0; A < 10; A++) acc = xq_add64(acc, (int64_t)vs[-VROW * A + 64 * (A & 1)] * coef[A]);
        dst[64 * s + lane] = (float)(int32_t)(acc >> 31) / 65536.0f;
      }
    }
    { /* state: ring blocks of the last 10 slots (1280 words over 128 threads), drc offset, window position */
      const int d_new = (d + RING - (32 * 128) % RING) % RING;
      int32_t tv[10];
#pragma unroll
      for (int j = 0; j < 10; j++) tv[j] = v[(9 + 32 - (1 + j)) * VROW + (int)threadIdx.x]; /* age A = 1 + j relative to the next frame's slot 0 */
#pragma unroll
      for (int j = 0; j < 10; j++) {
        int pos = d_new + 128 * (1 + j) + (int)threadIdx.x;
        if (pos >= RING) pos -= RING;
        if (pos >= RING) pos -= RING;
        st->ring[pos] = tv[j];
      }
      if (threadIdx.x == 0) {
        st->drc_offset = d_new;
        st->filt_off = (f_old + 32 * 64) % 640;
      }
    }
    __syncthreads(); /* channel 1's samples take the place of channel 0's */
  }
}

/* The two hand-offs around the Path A branch (decoder/ixheaacd_api.c:3385-3432, decoder/ixheaacd_decode_main.c:82-107): the
   core decoder's 16-bit PCM (channels interleaved) as floats, one plane per channel; and the branch's float output as 16-bit
   PCM, saturated and truncated towards zero, two channels interleaved.  One thread per output sample pair / sample. */
__global__ __launch_bounds__(256) void xaac_esbr_core_from_pcm16_kernel(XaacEsbrCoreInParams p) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; /* element of pcm: ((element * 1024 + k) * ch_fac + c) */
  if (e >= (size_t)p.n_ch * 1024) return;
  const int cf = p.ch_fac;
  const size_t el = e / ((size_t)1024 * cf), r = e % ((size_t)1024 * cf);
  const int k = (int)(r / cf), c = (int)(r % cf);
  p.core[(el * cf + c) * 1024 + k] = (float)p.pcm[e];
}
__global__ __launch_bounds__(256) void xaac_esbr_pcm16_from_float_kernel(XaacEsbrPcmOutParams p) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; /* (stream, sample) */
  if (e >= (size_t)p.n * 2048) return;
  const size_t i = e >> 11, k = e & 2047;
  const auto sat = [](float v) { /* decode_main.c:96-104 */
    if (v > 32767.0f) v = 32767.0f;
    else if (v < -32768.0f) v = -32768.0f;
    return (int16_t)v;
  };
  short2 o;
  o.x = sat(p.left[i * p.stride + k]);
  o.y = sat(p.right[i * p.stride + k]);
  reinterpret_cast<short2 *>(p.pcm)[e] = o;
}

/* The down-sampled synthesis bank of the eSBR branch (-dsample:1, output rates above 48 kHz: 32 synthesis channels, sbr_dec.c:556-569,
   :605-628; ixheaacd_esbr_qmfsyn32_winadd generic:1577): 32 samples a slot, the ring's first 640 words.  Lane = slot, two channel-frames
   a wave, as in the 24-channel analysis bank above and with its way of finding a ring block's words at a given slot: the slot's own 64
   new words, an earlier slot's of this frame, or the words the state holds.  Not a headline kernel: exact, short, untuned. */
__global__ __launch_bounds__(64) void xaac_esbr_synthesis_ds_kernel(XaacEsbrSynParams p) {
  constexpr int RS = 65, TS = 33;
  __shared__ int32_t old_ring[2][640];
  __shared__ int32_t blk[2][32 * RS]; /* slot k's 64 ring words */
  __shared__ int32_t win_lds[1280];
  __shared__ float tile[64 * TS];
  const int lane = threadIdx.x, lc = lane >> 5, slot = lane & 31;
  for (int i = lane; i < 1280; i += 64) win_lds[i] = xaac_qmf_esbr_qmf_c[i];
  int pb_c[2] = {0, 0}, fl_c[2] = {0, 0};
  bool live_c[2] = {false, false};
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const int ch = 2 * (int)blockIdx.x + c;
    if (ch >= p.n_ch) continue;
    if (p.only_ps && __builtin_amdgcn_readfirstlane((int)p.only_ps[ch].channel_mode) != 3) continue; /* no right channel in this frame */
    live_c[c] = true;
    const xaac_esbr_syn_state *st = reinterpret_cast<const xaac_esbr_syn_state *>(reinterpret_cast<const char *>(p.state) + (size_t)ch * p.state_stride);
    int d0 = st->drc_offset, fl0 = st->filt_off;
    for (int i = lane; i < 640; i += 64) old_ring[c][i] = st->ring[i];
    d0 = __builtin_amdgcn_readfirstlane(d0);
    fl0 = __builtin_amdgcn_readfirstlane(fl0);
    /* positions no run of the reference produces are brought onto the grids (blocks of 64 words, window steps of 64) */
    pb_c[c] = d0 < 0 ? 0 : (d0 >= 640 ? 9 : d0 / 64);
    fl_c[c] = fl0 < 0 ? 0 : (fl0 >= 640 ? 576 : fl0 / 64 * 64);
  }
  const int my_ch = 2 * (int)blockIdx.x + lc;
  const bool live = lc ? live_c[1] : live_c[0];
  const int pb = lc ? pb_c[1] : pb_c[0], fl0 = lc ? fl_c[1] : fl_c[0];
  if (live) { /* the slot's transform: rows in as (WORD32)(x * 64) (sbr_dec.c:592-595), its 64 ring words out */
    const float *re = p.qmf_re + (size_t)my_ch * p.in_stride + (size_t)slot * 64, *im = p.qmf_im + (size_t)my_ch * p.in_stride + (size_t)slot * 64;
    int32_t x[128], t[128], b[64];
#pragma unroll
    for (int k = 0; k < 32; k++) {
      x[k] = fx_f2i_trunc(re[k] * 64.0f);
      x[64 + k] = fx_f2i_trunc(im[k] * 64.0f);
    }
    xq_esbr_synth_slot_ds(x, t, b, 5 + 1);
#pragma unroll
    for (int k = 0; k < 64; k++) blk[lc][RS * slot + k] = b[k];
  }
  __syncthreads();
  /* block B of the ring as it stands at slot s: slot k wrote it if (pb - k) mod 10 == B; the newest such k <= s, or the state's words */
  const auto block_at = [&](int c, int pbc, int B, int s) -> const int32_t * {
    int k0 = pbc - B;
    k0 += k0 < 0 ? 10 : 0;
    if (k0 > s) return old_ring[c] + 64 * B;
    return blk[c] + RS * (k0 + 10 * ((s - k0) / 10));
  };
  if (live) { /* window-add: tmp1 = ring + f1, tmp2 = ring + f2, f1 0 | 32 by the slot's parity (sbr_dec.c:621-623) */
    const int f1 = (slot & 1) ? 32 : 0, f2 = 32 - f1;
    int fl = fl0 + 64 * slot;
    fl -= 640 * (fl / 640);
    const int32_t *bp[10];
#pragma unroll
    for (int B = 0; B < 10; B++) bp[B] = block_at(lc, pb, B, slot);
#pragma unroll 4
    for (int k = 0; k < 32; k++) {
      int64_t acc = 0;
#pragma unroll
      for (int j = 0; j < 5; j++) acc = xq_add64(acc, (int64_t)bp[2 * j][f1 + k] * win_lds[fl + 2 * (k + 64 * j)]);
#pragma unroll
      for (int j = 0; j < 5; j++) acc = xq_add64(acc, (int64_t)bp[2 * j + 1][f2 + k] * win_lds[fl + 2 * (k + 32 + 64 * j)]);
      tile[TS * lane + k] = (float)(int32_t)(acc >> 31) / 65536.0f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 2; c++) {
    if (!live_c[c]) continue;
    const int ch = 2 * (int)blockIdx.x + c;
    float *dst = p.out + (size_t)ch * (p.out_stride ? p.out_stride : 1024);
    for (int i = lane; i < 1024; i += 64) dst[i] = tile[TS * (32 * c + (i >> 5)) + (i & 31)];
    xaac_esbr_syn_state *st = reinterpret_cast<xaac_esbr_syn_state *>(reinterpret_cast<char *>(p.state) + (size_t)ch * p.state_stride);
    for (int i = lane; i < 640; i += 64) st->ring[i] = block_at(c, pb_c[c], i >> 6, 31)[i & 63];
    if (lane == 0) {
      int d = (pb_c[c] - 32) % 10;
      d += d < 0 ? 10 : 0;
      st->drc_offset = 64 * d;
      st->filt_off = (fl_c[c] + 64 * 32) % 640;
    }
  }
}

extern "C" hipError_t xaac_launch_esbr_synthesis_ds(const XaacEsbrSynParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_esbr_synthesis_ds_kernel, dim3((p->n_ch + 1) / 2), dim3(64), 0, stream, *p);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_esbr_core_from_pcm16(const XaacEsbrCoreInParams *p, hipStream_t stream) {
  const size_t total = (size_t)p->n_ch * 1024;
  hipLaunchKernelGGL(xaac_esbr_core_from_pcm16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, *p);
  return hipGetLastError();
}
extern "C" hipError_t xaac_launch_esbr_pcm16_from_float(const XaacEsbrPcmOutParams *p, hipStream_t stream) {
  const size_t total = (size_t)p->n * 2048;
  hipLaunchKernelGGL(xaac_esbr_pcm16_from_float_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, *p);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_esbr_analysis(const XaacEsbrAnaParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_esbr_analysis_kernel, dim3((p->n_ch + 1) / 2), dim3(64), XAAC_ESBR_ANA_LDS, stream, *p);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_esbr_analysis_nb(const XaacEsbrAnaNbParams *p, hipStream_t stream) {
  if (p->nb == 24) hipLaunchKernelGGL(xaac_esbr_analysis_nb_kernel<24>, dim3((p->n_ch + 1) / 2), dim3(64), 0, stream, *p); /* two channel-frames a wave */
  else hipLaunchKernelGGL(xaac_esbr_analysis_nb_kernel<16>, dim3(p->n_ch), dim3(64), 0, stream, *p);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_esbr_synthesis(const XaacEsbrSynParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_esbr_synthesis_kernel, dim3((p->n_ch + 1) / 2), dim3(128), XAAC_ESBR_SYN_LDS, stream, *p);
  return hipGetLastError();
}

/* xaac_warm_up (xaac_abi.cpp): asking for a kernel's attributes puts this translation unit's code object on the device */
extern "C" hipError_t xaac_warm_esbr_qmf(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&xaac_esbr_synthesis_kernel));
}
