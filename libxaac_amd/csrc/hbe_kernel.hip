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
 * work arrays private), and the windowed sums -- 32 x synth_size outputs of ten products
 * each, 16 x 4 synth_size of five -- are spread over all lanes with coalesced stores.  The delay lines are rewritten
 * once per frame from the last columns instead of shifted per column.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef XE_PROFILE /* stage timers of the DFT transposer's hops (tools/prof_hbe_dft.py): thread 0's cycles into xh_prof_total[0..5] */
extern __device__ unsigned long long xh_prof_total[8];
__shared__ long long xd_prof_last;
#define XD_T(i)                                                                   \
  do {                                                                            \
    if (threadIdx.x == 0) {                                                       \
      const long long t_ = clock64();                                             \
      atomicAdd(&xh_prof_total[i], (unsigned long long)(t_ - xd_prof_last));      \
      xd_prof_last = t_;                                                          \
    }                                                                             \
  } while (0)
#endif
#include "hbe_trans.h"
#include "hbe_dft.h"
#include "hbe_kernel.h"

#ifdef XE_PROFILE /* tools/prof_hbe_post.py: thread 0's cycles between the hooks of the products kernel, summed over channels */
__device__ unsigned long long xh_prof_total[8];
#define XH_T(i)                                                       \
  do {                                                                \
    if (threadIdx.x == 0) {                                           \
      const long long t_ = clock64();                                 \
      atomicAdd(&xh_prof_total[i], (unsigned long long)(t_ - xh_t0)); \
      xh_t0 = t_;                                                     \
    }                                                                 \
  } while (0)
extern "C" hipError_t xaac_debug_hbe_prof(unsigned long long *out, int clear) {
  if (clear) {
    unsigned long long z[8] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(xh_prof_total), z, sizeof(z));
  }
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(xh_prof_total), 8 * sizeof(unsigned long long));
}
#else
#define XH_T(i)
#endif

namespace {
/* the frame's pitch and whether the channel takes part (inside the Path A chain both come from the side info / frame) */
__device__ __forceinline__ int hbe_pitch(const int32_t *pitch, const xaac_esbr_side *side, int ch) {
  return side ? side[ch].pitch_in_bins : (pitch ? pitch[ch] : 0);
}
__device__ __forceinline__ bool hbe_skip(const xaac_sbr_frame *frame, int ch) { return frame && !frame[ch].apply_processing; }
}  // namespace

namespace {
struct XhWaveTeam { /* the channel's workgroup as the team of hbe_poly.h's cooperative routines */
  int lane, n;
  __device__ __forceinline__ void sync() const { __syncthreads(); }
};
}  // namespace

/* Both banks of one channel-frame in one launch (phases: XAAC_HBE_PHASE_SYNTH | XAAC_HBE_PHASE_ANAL): the synthesis bank's
   time signal stays in LDS for the analysis bank behind it (and goes to the state, which keeps it for the next frame).
   The columns' transforms run as team routines over LDS arrays -- (column, butterfly) units spread over the lanes,
   a barrier per pass -- instead of one column per lane in private (scratch) memory. */
/* (the bank size as a template parameter: every index split below divides by it or its double) */
template <int S>
__device__ __forceinline__ void xaac_hbe_banks_body(const XaacHbeBanksParams &p, xaac_hbe_state *st, int ch, int lane, float *lds) {
  float *T = lds;                 /* [(32 + 1) * s <= 660]: input_buf as the analysis bank reads it */
  float *R = lds + 33 * S; /* (the layout is the bank size's own: XAAC_HBE_BANKS_LDS_FLOATS(S)) */
  const bool synth = (p.phases & XAAC_HBE_PHASE_SYNTH) != 0, anal = (p.phases & XAAC_HBE_PHASE_ANAL) != 0;
  constexpr int s = S, a = 2 * S;
  const int ks = st->k_start, nc = p.num_columns;
  constexpr int NT = XAAC_HBE_BANKS_THREADS;
  const XhWaveTeam cx = {lane, NT};
  if (synth) {
    float(*vv)[2 * S] = reinterpret_cast<float(*)[2 * S]>(R); /* [9 + 32][2 s] */
    float *xin = R + 41 * 2 * S;                               /* [32][s] */
    float *work = xin + 32 * S;                                /* [32][4 s], [32][96] for the 24-point transforms */
    /* apply mode, hbe_trans.c:235-248: the last synth_size samples of the previous frame's time signal move to the
       front; while the reference's FFT pointers are unset it re-initialises, which clears both delay lines */
    const bool cleared = p.apply && !st->fft_ready;
    float front = 0.0f;
    if (lane < s) front = st->input_buf[p.apply ? nc * s + lane : lane];
    if (cleared) {
      for (int e = lane; e < 640; e += NT) st->analy_buf[e] = 0.0f;
      for (int e = 20 * s + lane; e < 1280; e += NT) st->synth_buf[e] = 0.0f;
    }
    for (int e = lane; e < 9 * 2 * s; e += NT) {
      const int c = -1 - e / (2 * s), t = e % (2 * s);
      vv[c + 9][t] = cleared ? 0.0f : xh_synth_hist(st->synth_buf, s, c, t);
    }
    const float *qre = p.qmf_re + (size_t)ch * p.in_stride, *qim = p.qmf_im + (size_t)ch * p.in_stride;
    for (int e = lane; e < nc * s; e += NT) {
      const int c = e / s, k = e % s;
      xin[S * c + k] = xh_synth_xin(qre + 64 * c, qim + 64 * c, ks, k);
    }
    __syncthreads();
    xh_synth_team(cx, [&](int c, int k) { return xin[S * c + k]; }, [&](int c) { return vv[c + 9]; }, nc, s, work);
    const auto at = [&](int c, int t) { return vv[c + 9][t]; };
    for (int o = lane; o < nc * s; o += NT) {
      const float y = xh_synth_out(at, s, o / s, o % s);
      st->input_buf[s + o] = y;
      T[s + o] = y;
    }
    if (lane < s) {
      if (p.apply) st->input_buf[lane] = front;
      T[lane] = front;
    }
    for (int e = lane; e < 20 * s; e += NT) {
      const int c = nc - 1 - e / (2 * s);
      st->synth_buf[e] = vv[c + 9][e % (2 * s)]; /* nc >= 1: columns nc - 10 .. nc - 1 >= -9 */
    }
    __syncthreads(); /* T complete; the clears of analy_buf above visible to the lanes that read it below */
  }
  if (!anal) return;
  constexpr int NCOL = XAAC_HBE_NO_BINS / 2;
  float(*u)[2 * a] = reinterpret_cast<float(*)[2 * a]>(R);                  /* [16][2 a] */
  float(*res)[2 * a] = reinterpret_cast<float(*)[2 * a]>(R + 16 * 2 * a);    /* [16][2 a] */
  float *work = R + 2 * 16 * 2 * a;                                          /* [16][4 a], [16][192] for the 48-point transforms */
  if (!synth) {
    for (int e = lane; e <= NCOL * a; e += NT) T[e] = st->input_buf[e];
    __syncthreads();
  }
  if (p.apply) /* hbe_trans.c:254-258: rows 16..27 become rows 0..11 (rows 12..27 are written below) */
    for (int e = lane; e < (XAAC_HBE_OPER_WIN_LEN - 1) * 128; e += NT) st->qmf_in_buf[e >> 7][e & 127] = st->qmf_in_buf[(e >> 7) + NCOL][e & 127];
  for (int e = lane; e < NCOL * 2 * a; e += NT) u[e / (2 * a)][e % (2 * a)] = xh_anal_u(T, st->analy_buf, a, e / (2 * a), e % (2 * a));
  /* the delay line the last column leaves (read before anything of it is overwritten) */
  constexpr int NB = (10 * a + NT - 1) / NT;
  float nb[NB];
#pragma unroll
  for (int q = 0; q < NB; q++) {
    const int n = lane + NT * q;
    nb[q] = n < 10 * a ? xh_anal_x(T, st->analy_buf, a, NCOL - 1, n) : 0.0f;
  }
  __syncthreads();
  xh_anal_team(cx, &u[0][0], 2 * a, [&](int c) { return res[c]; }, NCOL, a, work);
  for (int e = lane; e < NCOL * 128; e += NT) {
    const int idx = e >> 7, w = (e & 127) - 4 * ks;
    st->qmf_in_buf[idx + XAAC_HBE_OPER_WIN_LEN - 1][e & 127] = (w >= 0 && w < 2 * a) ? res[idx][w] : 0.0f;
  }
#pragma unroll
  for (int q = 0; q < NB; q++) {
    const int n = lane + NT * q;
    if (n < 10 * a) st->analy_buf[n] = nb[q];
  }
}

__global__ __launch_bounds__(XAAC_HBE_BANKS_THREADS) void xaac_hbe_banks_kernel(XaacHbeBanksParams p) {
  extern __shared__ float lds[];
  const int ch = blockIdx.x, lane = threadIdx.x;
  xaac_hbe_state *st = p.state + ch;
  if (hbe_skip(p.frame, ch)) return;
  const bool synth = (p.phases & XAAC_HBE_PHASE_SYNTH) != 0, anal = (p.phases & XAAC_HBE_PHASE_ANAL) != 0;
  const int s = st->synth_size, ks = st->k_start, nc = p.num_columns, a = 2 * s;
  bool bad;
  if (p.apply) bad = !xh_apply_params_ok(st, hbe_pitch(p.pitch, p.side, ch));
  else
    bad = !xh_size_ok(s) || ks < 0 || (synth && (ks + s > 64 || ks * 32 + 2 * s > 7 * 64 || nc < 0 || nc > 32)) ||
          (anal && 4 * ks + 2 * a > 128);
  /* the launch's LDS was sized for p.lds_synth_size (the largest bank the host expects; 0: any): a larger bank is refused */
  bad = bad || !XAAC_HBE_LDS_OK(s, p.lds_synth_size);
  if (lane == 0 && p.status && (synth || !p.apply)) p.status[ch] = bad ? -1 : 0;
  if (bad) return;
  switch (s) {
    case 4: xaac_hbe_banks_body<4>(p, st, ch, lane, lds); break;
    case 8: xaac_hbe_banks_body<8>(p, st, ch, lane, lds); break;
    case 12: xaac_hbe_banks_body<12>(p, st, ch, lane, lds); break;
    case 16: xaac_hbe_banks_body<16>(p, st, ch, lane, lds); break;
    default: xaac_hbe_banks_body<20>(p, st, ch, lane, lds); break;
  }
}

/* 256 threads per channel.  Tiles of 16 output bands.  Per tile: (1) the normalised samples the tile's products read --
   a function of (row, input band) alone, each used by up to ten columns -- are computed once into LDS (five planes:
   fourth-root, 3/4-power and cube-root normalisations, the cube-root-normalised interpolated points in their two
   summation orders); (2) thread (band, column) forms the column's block of products (+ cross terms) in LDS; (3) thread
   per complex element (row, band) moves the previous frame's upper rows down (rows 0..31 first, the rows they come from
   afterwards), adds the blocks that reach it in column order, and rotates rows 0..31 of the SBR range into the output
   (hbe_trans.c:262-295). */
namespace {
enum { HP_N2, HP_N4, HP_N3A, HP_N3B1, HP_N3B2, HP_PLANES };
constexpr int HP_SLOTS = 16; /* input bands per plane and tile (plane_base .. + 15) */
__device__ __forceinline__ int hp_base(int plane, int tile) { /* the lowest input band a tile's products can read */
  return plane == HP_N2 ? 16 * tile : (plane == HP_N4 ? (8 * tile > 0 ? 8 * tile - 1 : 0) : (32 * tile) / 3);
}
}  // namespace

__global__ __launch_bounds__(XAAC_HBE_POST_THREADS) void xaac_hbe_post_kernel(XaacHbePostParams p) {
  extern __shared__ float lds[];
  /* [16 bands x 16 columns]: xh_column_block of (band bt, column i) in row 17 bt + i -- the skew spreads the gather's reads,
     whose lanes differ in bt, over the LDS banks (rows 16 bt + i put all sixteen bands on two banks) */
  float(*blk)[XH_BLK] = reinterpret_cast<float(*)[XH_BLK]>(lds);
  float2(*nv)[HP_SLOTS][32] = reinterpret_cast<float2(*)[HP_SLOTS][32]>(lds + XAAC_HBE_POST_BLK_ROWS * XH_BLK); /* [plane][slot][row] */
  __shared__ unsigned need[HP_PLANES]; /* bit s: slot s of the plane is read by this tile */
  const int ch = blockIdx.x, tid = threadIdx.x;
  xaac_hbe_state *st = p.state + ch;
  if (hbe_skip(p.frame, ch)) return;
  const int pitch = hbe_pitch(p.pitch, p.side, ch);
  if (!xh_apply_params_ok(st, pitch) || !XAAC_HBE_LDS_OK(st->synth_size, p.lds_synth_size)) return;
  const int ms = st->max_stretch, sb0 = st->start_band, sb1 = st->end_band;
  int32_t xo[4];
  for (int q = 0; q < 4; q++) xo[q] = st->x_over_qmf[q];
  float *pv_re = p.pv_re + (size_t)ch * p.pv_stride, *pv_im = p.pv_im + (size_t)ch * p.pv_stride;
  const float *in_flat = &st->qmf_in_buf[0][0];
  const auto inf = [&](int row, int idx) { return in_flat[128 * row + idx]; };
#ifdef XE_PROFILE
  long long xh_t0 = clock64();
#endif
  for (int tile = 0; tile < 4; tile++) {
    /* the previous frame's upper rows of this tile's bands -- the start values of rows 0..31 in (3) -- are on their way
       from memory while (1) and (2) compute */
    float2 upper[(32 * 16) / XAAC_HBE_POST_THREADS];
#pragma unroll
    for (int q = 0; q < (32 * 16) / XAAC_HBE_POST_THREADS; q++) {
      const int e = tid + q * XAAC_HBE_POST_THREADS;
      upper[q] = *reinterpret_cast<const float2 *>(&st->qmf_out_buf[32 + (e >> 4)][2 * (16 * tile + (e & 15))]);
    }
    /* (1) which normalised samples the tile reads, then those samples */
    if (tid < HP_PLANES) need[tid] = 0;
    __syncthreads();
    if (tid < 16) {
      const int qb = 16 * tile + tid, f = xh_band_factor(xo, ms, qb);
      if (f == 2) {
        atomicOr(&need[HP_N2], 1u << (qb - hp_base(HP_N2, tile)));
      } else if (f == 4) {
        const int inp = qb >> 1, ip = (qb & 1) ? inp + 1 : inp - 1, b = hp_base(HP_N4, tile);
        atomicOr(&need[HP_N4], (1u << (inp - b)) | (1u << (ip - b)));
      } else if (f == 3) {
        const int inp = (2 * qb) / 3, rem = 2 * qb - 3 * inp, b = hp_base(HP_N3A, tile);
        if (rem == 2) {
          atomicOr(&need[HP_N3A], 3u << (inp - b));
          atomicOr(&need[HP_N3B2], 3u << (inp - b));
        } else {
          atomicOr(&need[HP_N3A], 1u << (inp - b));
          atomicOr(&need[HP_N3B1], 1u << (inp - b));
        }
      }
    }
    __syncthreads();
    /* thread's entries e = tid + 256 q: two per plane (q >> 1); every input sample they read is requested before the
       first one is used (one memory latency for the phase instead of one per entry) */
    static_assert(HP_SLOTS * 32 == 2 * XAAC_HBE_POST_THREADS, "two entries per plane and thread");
    float2 xa[2 * HP_PLANES], xb[4];
    bool on[2 * HP_PLANES];
#pragma unroll
    for (int q = 0; q < 2 * HP_PLANES; q++) {
      const int plane = q >> 1, e = tid + (q & 1) * XAAC_HBE_POST_THREADS, slot = e >> 5, row = e & 31;
      const int band = hp_base(plane, tile) + slot;
      on[q] = ((need[plane] >> slot) & 1) != 0 && (plane < HP_N3B1 || row < 30);
      xa[q] = make_float2(0.0f, 0.0f);
      if (plane >= HP_N3B1) xb[q - 2 * HP_N3B1] = make_float2(0.0f, 0.0f);
      if (on[q]) {
        if (plane < HP_N3B1) {
          xa[q] = *reinterpret_cast<const float2 *>(&st->qmf_in_buf[row][2 * band]);
        } else { /* the interpolated point behind `row` reads rows row + 1, row + 2 (used: row <= 24) */
          xa[q] = *reinterpret_cast<const float2 *>(&st->qmf_in_buf[row + 2][2 * band]);
          xb[q - 2 * HP_N3B1] = *reinterpret_cast<const float2 *>(&st->qmf_in_buf[row + 1][2 * band]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 2 * HP_PLANES; q++) {
      const int plane = q >> 1, e = tid + (q & 1) * XAAC_HBE_POST_THREADS, slot = e >> 5, row = e & 31;
      if (!((need[plane] >> slot) & 1)) continue;
      const int band = hp_base(plane, tile) + slot;
      XhC v = {0.0f, 0.0f};
      if (plane == HP_N2) {
        v = xh_norm2(xa[q].x, xa[q].y);
      } else if (plane == HP_N4) {
        v = xh_norm4(xa[q].x, xa[q].y);
      } else if (plane == HP_N3A) {
        v = xh_norm3(xa[q].x, xa[q].y);
      } else if (on[q]) {
        const float2 r2 = xa[q], r1 = xb[q - 2 * HP_N3B1 < 0 ? 0 : q - 2 * HP_N3B1];
        const XhC x = xh_interp3([&](int rr, int) { const XhC c = rr == row + 2 ? XhC{r2.x, r2.y} : XhC{r1.x, r1.y}; return c; }, band, row,
                                 plane == HP_N3B2);
        v = xh_norm3(x.r, x.i);
      }
      nv[plane][slot][row] = make_float2(v.r, v.i);
    }
    __syncthreads();
    XH_T(0);
    /* (2) the columns' blocks */
    {
      const int qb = 16 * tile + (tid >> 4), i = tid & 15;
      float *mine = blk[17 * (tid >> 4) + i];
      const int f = xh_band_factor(xo, ms, qb);
      const auto cached = [&](int plane, int band, int row) {
        const float2 v = nv[plane][band - hp_base(plane, tile)][row];
        const XhC c = {v.x, v.y};
        return c;
      };
      if (f == 2) {
        xh_prod2_block_n([&](int row) { return cached(HP_N2, qb, row); }, i, mine);
      } else if (f == 4) {
        const int inp = qb >> 1, ip = (qb & 1) ? inp + 1 : inp - 1;
        xh_prod4_block_n([&](int row) { return cached(HP_N4, inp, row); }, [&](int row) { return cached(HP_N4, ip, row); }, i, mine);
      } else if (f == 3) {
        const int inp = (2 * qb) / 3, rem = 2 * qb - 3 * inp;
        xh_prod3_block_n([&](int sel, int row) { return cached(HP_N3A, inp + sel, row); },
                         [&](int sel, int row) { return cached(rem == 2 ? HP_N3B2 : HP_N3B1, inp + sel, row); }, rem, i, mine);
      }
      if (f) xh_column_cross(inf, f, qb, i, pitch, mine);
    }
    __syncthreads();
    XH_T(1);
    /* (3) rows r and r + 32 of the thread's band: the old upper row (fetched above) moves down and takes the blocks that
       reach it, the new upper row starts from zero; the lower row is rotated into the output */
#pragma unroll
    for (int q = 0; q < (32 * 16) / XAAC_HBE_POST_THREADS; q++) {
      const int e = tid + q * XAAC_HBE_POST_THREADS;
      const int r = e >> 4, bt = e & 15, qb = 16 * tile + bt;
      float2 lo = upper[q], hi = make_float2(0.0f, 0.0f);
      const int f = xh_band_factor(xo, ms, qb);
      if (f) {
        const auto bk = [&](int i) { return (const float *)blk[17 * bt + i]; };
        xh_prod_gather2(lo.x, lo.y, f, r, bk);
        xh_prod_gather2(hi.x, hi.y, f, r + 32, bk);
      }
      *reinterpret_cast<float2 *>(&st->qmf_out_buf[r][2 * qb]) = lo;
      *reinterpret_cast<float2 *>(&st->qmf_out_buf[r + 32][2 * qb]) = hi;
      if (qb >= sb0 && qb < sb1) {
        pv_re[64 * r + qb] = (float)(lo.x * xaac_hbe_pv_cos[qb] - lo.y * xaac_hbe_pv_sin[qb]);
        pv_im[64 * r + qb] = (float)(lo.x * xaac_hbe_pv_sin[qb] + lo.y * xaac_hbe_pv_cos[qb]);
      } else if (p.zero_outside) {
        pv_re[64 * r + qb] = 0.0f;
        pv_im[64 * r + qb] = 0.0f;
      }
    }
    __syncthreads(); /* the next tile rewrites the planes and the blocks */
    XH_T(2);
  }
  if (tid == 0 && !st->fft_ready && st->synth_size != 20) st->fft_ready = 1;
}

/* The DFT transposer's analysis bank: 256 threads per channel; u of all columns into LDS (five window products each), then
   thread = (column, sub-band) runs the 2 L-term sums against the coefficient matrices (global, shared by the channels of
   a configuration), and the rows' cleared cells are written in the closed form of the reference's overlapping memsets. */
__global__ __launch_bounds__(256) void xaac_hbe_dft_anal_kernel(XaacHbeDftParams p) {
  extern __shared__ float lds[];
  float(*u)[128] = reinterpret_cast<float(*)[128]>(lds);
  const int ch = blockIdx.x, tid = threadIdx.x, nb = p.no_bins;
  if (p.chain && *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(p.status_in) + (size_t)ch * p.status_stride) != 0) return;
  if (hbe_skip(p.frame, ch)) return;
  xaac_hbe_dft_anal_state *st = reinterpret_cast<xaac_hbe_dft_anal_state *>(
      reinterpret_cast<char *>(p.state) + (size_t)ch * (p.state_stride ? (size_t)p.state_stride : sizeof(xaac_hbe_dft_anal_state)));
  const int L = st->analy_size, a0 = st->a_start;
  const bool bad = L < 4 || L > 64 || (L & 3) || a0 < 0 || a0 + L > 64 || nb < 1 || nb > 32;
  if (tid == 0 && p.status && !p.chain) p.status[ch] = bad ? -1 : 0;
  if (bad) return;
  const float *tin = p.time_in + (size_t)ch * p.in_stride, *win = xh_window_dft(L);
  const size_t c = p.cfg ? (size_t)p.cfg[ch] : 0;
  const float *cre = p.coef_re + c * 64 * 128, *cim = p.coef_im + c * 64 * 128;
  const size_t qs = p.qmf_stride ? (size_t)p.qmf_stride : (size_t)(nb + 2) * 64;
  const int rows = p.max_rows ? p.max_rows : nb + 2;
  float *qre = p.qmf_re + (size_t)ch * qs, *qim = p.qmf_im + (size_t)ch * qs;
  for (int e = tid; e < nb * 2 * L; e += 256) u[e / (2 * L)][e % (2 * L)] = xh_anal_u_w(tin, st->analy_buf, L, e / (2 * L), e % (2 * L), win);
  float keep[3];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const int n = tid + 256 * q;
    keep[q] = n < 10 * L ? xh_anal_x(tin, st->analy_buf, L, nb - 1, n) : 0.0f;
  }
  __syncthreads();
  for (int e = tid; e < rows * 64; e += 256) {
    const int idx = e >> 6, k = e & 63;
    if (p.zero_below && idx < nb && k < a0) {
      qre[e] = 0.0f;
      qim[e] = 0.0f;
      continue;
    }
    if (idx < nb && k >= a0 && k < a0 + L) {
      float o_r, o_i;
      xh_dft_anal_band(u[idx], 2 * L, cre + 128 * (k - a0), cim + 128 * (k - a0), o_r, o_i);
      qre[e] = o_r;
      qim[e] = o_i;
    } else {
      if (idx < nb && k >= a0) qre[e] = 0.0f;
      if ((idx < nb && (idx > 0 || k >= a0)) || idx == nb || (idx == nb + 1 && k < a0)) qim[e] = 0.0f;
    }
  }
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const int n = tid + 256 * q;
    if (n < 10 * L) st->analy_buf[n] = keep[q];
  }
}

/* ixheaacd_dft_hbe_apply up to its output signal (hbe_dft_trans.c:771-937): the signals' shifts, the real synthesis bank in
   its esbr_hq layout (esbr_polyphase.c:170-182: column idx's samples at ana_fft_size[0] + (idx - 1) synth_size), the eight
   hops.  One workgroup per channel-frame; every loop of hbe_dft.h spreads over the 256 lanes with a barrier per phase.  The
   analysis bank (xaac_hbe_dft_anal_kernel, chain mode) follows as its own launch on the output signal in the state. */
__global__ __launch_bounds__(XAAC_HBE_DFT_CORE_THREADS) void xaac_hbe_dft_core_kernel(XaacHbeDftCoreParams p) {
  extern __shared__ float lds[];
  const int ch = blockIdx.x, lane = threadIdx.x;
  constexpr int NT = XAAC_HBE_DFT_CORE_THREADS;
  xaac_hbe_dft_state *st = p.state + ch;
  if (hbe_skip(p.frame, ch)) {
    if (lane == 0) st->last_status = 1;
    return;
  }
  const int ovs = p.side ? ((p.side[ch].harmonic_sbr & XAAC_ESBR_OVERSAMPLING) ? 1 : 0) : (p.oversampling && p.oversampling[ch] ? 1 : 0);
  const int pitch = hbe_pitch(p.pitch, p.side, ch);
  const xaac_hbe_dft_cfg *cfg = p.cfg_tab + (p.cfg ? p.cfg[ch] : 0);
#ifdef XE_PROFILE
  const long long xd_t_start = clock64();
#endif
  XdSizes z;
  const bool ok = xd_sizes(st, ovs, &z); /* (uniform: every lane reads the same words) */
  if (lane == 0) {
    if (p.status) p.status[ch] = ok ? 0 : -1;
    st->last_status = ok ? 0 : -1;
  }
  if (!ok) return;
  float *in = lds, *out = in + 2 * XAAC_HBE_DFT_MAX_ANA;
  XdC *wa = reinterpret_cast<XdC *>(out + 4 * XAAC_HBE_DFT_MAX_SYN), *ws = wa + 384;
  float *U = reinterpret_cast<float *>(ws + 384);
  const int s = z.s, ks = st->k_start;
  const XhWaveTeam cx = {lane, NT};
  /* :800-809: the signals move down by a frame; the output's upper half starts from zero */
  for (int e = lane; e < z.ana0; e += NT) in[e] = st->input_buf[z.ana0 + e];
  for (int e = 2 * z.ana0 - s + lane; e < 2 * z.ana0; e += NT) in[e] = st->input_buf[e]; /* (what column 32 would write: kept) */
  for (int e = lane; e < 2 * z.syn0; e += NT) {
    out[e] = st->output_buf[2 * z.syn0 + e];
    out[2 * z.syn0 + e] = 0.0f;
  }
  { /* the synthesis bank: as xaac_hbe_banks_body's first phase */
    float *vv = U;                 /* [9 + 32][2 s] */
    float *xin = vv + 41 * 2 * s;  /* [32][s] */
    float *work = xin + 32 * s;    /* [32][4 s], [32][96] for the 24-point transforms */
    for (int e = lane; e < 9 * 2 * s; e += NT) {
      const int c = -1 - e / (2 * s), t = e % (2 * s);
      vv[(c + 9) * 2 * s + t] = xh_synth_hist(st->synth_buf, s, c, t);
    }
    const float *qre = p.qmf_re + (size_t)ch * 2048, *qim = p.qmf_im + (size_t)ch * 2048;
    for (int e = lane; e < 32 * s; e += NT) {
      const int c = e / s, k = e % s;
      xin[e] = xh_synth_xin(qre + 64 * c, qim + 64 * c, ks, k);
    }
    __syncthreads();
    xh_synth_team(cx, [&](int c, int k) { return xin[s * c + k]; }, [&](int c) { return vv + (c + 9) * 2 * s; }, 32, s, work);
    const auto at = [&](int c, int t) { return vv[(c + 9) * 2 * s + t]; };
    for (int o = lane; o < 32 * s; o += NT) in[z.ana0 - s + o] = xh_synth_out(at, s, o / s, o % s);
    for (int e = lane; e < 20 * s; e += NT) st->synth_buf[e] = vv[(31 - e / (2 * s) + 9) * 2 * s + e % (2 * s)];
    __syncthreads(); /* vv is dead: the hops' arrays take its place */
  }
#ifdef XE_PROFILE
  if (threadIdx.x == 0) { /* [6]: everything in front of the hops (loads, the synthesis bank) */
    const long long t_ = clock64();
    atomicAdd(&xh_prof_total[6], (unsigned long long)(t_ - xd_t_start));
    xd_prof_last = t_;
  }
#endif
  XdWork w;
  w.in = in;
  w.out = out;
  w.spec = U;        /* 768 words */
  w.awin = U + 768;  /* 512 words (+ 256 spare: the hop's arrays keep the places the 1536-word spectrum gave them) */
  w.tx = U + 1536;
  w.mag = w.tx + 1540;
  w.phase = w.mag + 772;
  w.wa = wa;
  w.ws = ws;
  w.tmp = reinterpret_cast<XdC *>(w.phase + 772);
  xd_hops(cx, z, cfg, ovs, pitch, &w);
  for (int e = lane; e < 2 * z.ana0; e += NT) st->input_buf[e] = in[e];
  for (int e = lane; e < 4 * z.syn0; e += NT) st->output_buf[e] = out[e];
}

extern "C" hipError_t xaac_launch_hbe_dft_core(const XaacHbeDftCoreParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_hbe_dft_core_kernel, dim3(p->n_ch), dim3(XAAC_HBE_DFT_CORE_THREADS), XAAC_HBE_DFT_CORE_LDS, stream, *p);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_hbe_dft_anal(const XaacHbeDftParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_hbe_dft_anal_kernel, dim3(p->n_ch), dim3(256), XAAC_HBE_DFT_LDS, stream, *p);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_hbe_banks(const XaacHbeBanksParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_hbe_banks_kernel, dim3(p->n_ch), dim3(XAAC_HBE_BANKS_THREADS),
                     (size_t)XAAC_HBE_BANKS_LDS_FLOATS_FOR(p->lds_synth_size) * 4, stream, *p);
  return hipGetLastError();
}

extern "C" hipError_t xaac_launch_hbe_post(const XaacHbePostParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_hbe_post_kernel, dim3(p->n_ch), dim3(XAAC_HBE_POST_THREADS), XAAC_HBE_POST_LDS, stream, *p);
  return hipGetLastError();
}

/* xaac_warm_up (xaac_abi.cpp): asking for a kernel's attributes puts this translation unit's code object on the device */
extern "C" hipError_t xaac_warm_hbe(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&xaac_hbe_post_kernel));
}
