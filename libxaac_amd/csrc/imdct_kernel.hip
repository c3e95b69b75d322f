/*
 * imdct_kernel.hip -- gfx950 kernel for the AAC 1024-sample IMDCT + window /
 * overlap-add (+ fused PCM16 hand-off), one wave64 per channel-frame.
 *
 * Replaces, per channel-frame, the reference's ixheaacd_imdct_process
 * (decoder/ixheaacd_lpfuncs.c:347-802, frame_length 1024) and everything it
 * calls through the selector: calc_max_spectral_line (aac_tns.c:422),
 * pretwiddle_compute (aac_imdct.c:165), imdct_using_fft (:834), post_twiddle
 * (:331), post_twid_overlap_add (:506), process_win_seq / long_short_win_seq /
 * nolap1_32 / neg_shift_spec / spec_to_overlapbuf / overlap_buf_out /
 * overlap_out_copy (lpfuncs.c:94-345), over_lap_add1/2 (block.c:1193-1240) and
 * the WORD32->WORD16 hand-off (api.c:353-366 / peak_limiter.c:324 + api.c:3676).
 *
 * MI355X mapping (docs/NOTEBOOK.md §3-4):
 *   - persistent grid; a 256-thread workgroup = 4 independent waves, each wave
 *     loops over channel-frames (no __syncthreads in the loop, LDS regions are
 *     private to a wave, a wave's DS ops retire in order).
 *   - HBM: spec (4 KB) and overlap (2 KB) come in as 16 B/lane coalesced loads;
 *     PCM / overlap go out as 8-16 B/lane coalesced stores on the common path.
 *   - 512-point complex FFT = 3 radix-8 passes, ONE butterfly per lane per pass,
 *     data exchanged through a 4 KB LDS tile whose index swizzle
 *     phi(A,B,C) = 64A + 8(B ^ (A&3)) + (A ^ C) makes every pass's ds_read_b64 /
 *     ds_write_b64 bank-conflict free.
 *   - rotation + FFT twiddle factors live in VGPRs for the lifetime of the wave
 *     (pre-shifted so that every 32x16 multiply is a single v_mul_hi_i32);
 *     windows (4.5 KB) are staged in LDS once per workgroup.
 *   - integer only; no MFMA (fixed-size transforms, not a dense contraction).
 * Arithmetic is the reference's Q-format behaviour bit for bit (fx.h).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fx.h"
#define XAAC_TAB_QUAL static __device__ const
#include "tables_imdct.inc"
#include "imdct_kernel.h"

namespace {

/* LDS-qualified element types: the out-of-line rare paths take these so that they
   use ds_read/ds_write and not FLAT addressing (a flat base address that dips below
   the LDS aperture before its immediate offset is added faults on gfx950). */
typedef __attribute__((address_space(3))) int32_t lds_i32;
typedef __attribute__((address_space(3))) int16_t lds_i16;

struct cpx {
  int32_t re, im;
};
__device__ __forceinline__ cpx c_add(cpx a, cpx b) { return {fx_add(a.re, b.re), fx_add(a.im, b.im)}; }
__device__ __forceinline__ cpx c_sub(cpx a, cpx b) { return {fx_sub(a.re, b.re), fx_sub(a.im, b.im)}; }
__device__ __forceinline__ cpx c_sub_j(cpx a, cpx b) { return {fx_add(a.re, b.im), fx_sub(a.im, b.re)}; }
__device__ __forceinline__ cpx c_add_j(cpx a, cpx b) { return {fx_sub(a.re, b.im), fx_add(a.im, b.re)}; }

constexpr int32_t kSqrtHalfHi = (int32_t)(0x5A82u << 16); /* aac_imdct.c:971, pre-shifted for mulhi */

/* radix-8 butterfly, wrapping arithmetic; y[q] is the value stored q*del after
   the butterfly's base (aac_imdct.c:876-999 and its two twiddled variants,
   merged by ring identities mod 2^32 -- see oracle/oracle_imdct.c:xo_bfly8). */
__device__ __forceinline__ void bfly8(const cpx (&x)[8], cpx (&y)[8]) {
  cpx e0 = c_add(x[0], x[4]), e4 = c_sub(x[0], x[4]);
  cpx e2 = c_add(x[2], x[6]), e6 = c_sub(x[2], x[6]);
  cpx f0 = c_add(e0, e2), f2 = c_sub(e0, e2);
  cpx f4 = c_sub_j(e4, e6), f6 = c_add_j(e4, e6);
  cpx g1 = c_add(x[1], x[5]), g5 = c_sub(x[1], x[5]);
  cpx g3 = c_add(x[3], x[7]), g7 = c_sub(x[3], x[7]);
  cpx h1 = c_add(g1, g3), h3 = c_sub(g1, g3);
  int32_t s5 = fx_add(g5.re, g5.im), d5 = fx_sub(g5.re, g5.im);
  int32_t s7 = fx_add(g7.re, g7.im), d7 = fx_sub(g7.re, g7.im);
  int32_t p7i = fx_shlw(fx_sub(s5, d7), 1);
  int32_t p5r = fx_shlw(fx_neg(fx_add(s5, d7)), 1);
  int32_t p5i = fx_shlw(fx_sub(s7, d5), 1);
  int32_t p7r = fx_shlw(fx_neg(fx_add(s7, d5)), 1);
  cpx m7 = {fx_mulhi(p7i, kSqrtHalfHi), fx_mulhi(p7r, kSqrtHalfHi)};
  cpx m5 = {fx_mulhi(p5i, kSqrtHalfHi), fx_mulhi(p5r, kSqrtHalfHi)};
  y[0] = c_add(f0, h1);
  y[4] = c_sub(f0, h1);
  y[2] = c_sub_j(f2, h3);
  y[6] = c_add_j(f2, h3);
  y[1] = c_add(f4, m7);
  y[5] = c_sub(f4, m7);
  y[3] = c_add(f6, m5);
  y[7] = c_sub(f6, m5);
}

/* packed twiddle: hi16 = -sin, lo16 = cos; product doubled (aac_imdct.c:1179-1185) */
__device__ __forceinline__ cpx twiddle(cpx x, int32_t w) {
  int32_t wl = (int32_t)((uint32_t)w << 16), wh = (int32_t)((uint32_t)w & 0xffff0000u);
  cpx r;
  r.re = fx_shlw(fx_sub(fx_mulhi(x.re, wl), fx_mulhi(x.im, wh)), 1);
  r.im = fx_shlw(fx_add(fx_mulhi(x.re, wh), fx_mulhi(x.im, wl)), 1);
  return r;
}

/* LDS exchange-tile swizzle for logical complex index 64A + 8B + C */
__device__ __forceinline__ int phi(int A, int B, int C) { return (A << 6) | ((B ^ (A & 3)) << 3) | (A ^ C); }

/* rotation pair for bin c of an n2-bin transform (pre- and post-twiddle share it):
   aac_imdct.c:174-328 / :339-421; returned pre-shifted by 16 for mulhi */
__device__ __forceinline__ void rot_pair(int c, int n2, int st, int32_t &X, int32_t &Y) {
  int n4 = n2 >> 1;
  int16_t x, y;
  if (c <= n4) {
    int p = c * st;
    x = xaac_tab_pre_cs[2 * p];
    y = xaac_tab_pre_cs[2 * p + 1];
  } else {
    int p = (n2 - c) * st;
    x = xaac_tab_pre_cs[2 * p + 1];
    y = xaac_tab_pre_cs[2 * p];
  }
  X = (int32_t)((uint32_t)(uint16_t)x << 16);
  Y = (int32_t)((uint32_t)(uint16_t)y << 16);
}

/* pre-rotation scaling (aac_imdct.c:177-328): e < 0 -> wrapping left shift by -e,
   else arithmetic right shift by e.  e = 9 - headroom (long) or 6 - headroom
   (short) with headroom in [0,31], so both counts stay below 26 and the
   reference's "count > 31" guards can never fire: two plain shifts, one of them by 0. */
__device__ __forceinline__ int32_t scale_by_expo(int32_t v, int sl, int sr) { return fx_shlw(v, sl) >> sr; }

__device__ __forceinline__ int32_t wave_or(int32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
  return v;
}

/* maximum of a non-negative value over the wave, in the VALU's data-parallel-primitive lanes (no LDS round
   trips): butterfly inside each row of 16, then the row results ripple up; lane 63 holds the total */
__device__ __forceinline__ int32_t wave_max_nonneg(int32_t v) {
#define XAAC_DPP_MAX(ctrl, rows)                                                   \
  {                                                                                \
    const int32_t t_ = __builtin_amdgcn_update_dpp(v, v, ctrl, rows, 0xf, false);  \
    v = t_ > v ? t_ : v;                                                           \
  }
  XAAC_DPP_MAX(0xB1, 0xf)  /* quad_perm [1,0,3,2] */
  XAAC_DPP_MAX(0x4E, 0xf)  /* quad_perm [2,3,0,1] */
  XAAC_DPP_MAX(0x141, 0xf) /* row_half_mirror */
  XAAC_DPP_MAX(0x140, 0xf) /* row_mirror */
  XAAC_DPP_MAX(0x142, 0xa) /* row_bcast15 into rows 1, 3 */
  XAAC_DPP_MAX(0x143, 0xc) /* row_bcast31 into rows 2, 3 */
#undef XAAC_DPP_MAX
  return __builtin_amdgcn_readlane(v, 63);
}

/* 16-bit multiplier from an LDS / register window value */
__device__ __forceinline__ int32_t mul16(int32_t a, int16_t w) { return fx_mul32x16(a, w); }
__device__ __forceinline__ int32_t nosh(int32_t a, int16_t w) {
  /* full 48-bit product clamped to 32 bits, no shift (aac_imdct.c:95-106) */
  int64_t p = (int64_t)a * (int64_t)w;
  return fx_sat64(p);
}

/* where a finished time sample goes: WORD32 block and/or PCM16 after the qshift hand-off */
struct Sink {
  int32_t *o32;
  int16_t *p16;
  int stride;
  int qadj;
  int mode;
  /* The reference converts an interleaved stereo block for the SBR tool IN PLACE, channel by channel
     (ixheaacd_allocate_sbr_scr, api.c:353-366): channel 0's WORD16 results land in the low halves of words 0..1023,
     and the odd ones of those are channel 1's samples 0..511, not yet converted.  So channel 1's sample n < 512 is
     converted with its low 16 bits replaced by channel 0's PCM sample 2 n + 1.  peer: channel 0's PCM of this access
     unit (interleaved output, stride 2), set only for channel 1 of a stereo unit in XAAC_PCM_SBR mode. */
  const int16_t *peer;
  __device__ __forceinline__ int16_t to_pcm(int32_t v) const {
    return fx_round16(mode ? fx_shl_sat(v, qadj) : fx_shlw(v, qadj));
  }
  __device__ __forceinline__ int32_t in_place(int n, int32_t v) const {
    if (peer && n < 512) v = (int32_t)(((uint32_t)v & 0xffff0000u) | (uint16_t)peer[2 * (2 * n + 1)]);
    return v;
  }
  __device__ __forceinline__ void put(int n, int32_t v) const {
    if (o32) o32[n * stride] = v;
    if (p16) p16[n * stride] = to_pcm(in_place(n, v));
  }
};

/* ------------------------------------------------------------------------- */
/* Lane-major constant tiles in LDS (filled once per workgroup): every read is one 8-byte
   word [k*64 + lane] -> bank-conflict free, and keeps ~30 VGPRs free.
     rot[k][lane]    rotation pair (X, Y) of bin lane + 64k (n = 1024), pre-shifted << 16
     tw2[k-1][lane]  pass-2 twiddle of column m = lane & 7 : tw[8*m*k], split (cos << 16, -sin << 16)
     tw3[k-1][lane]  pass-3 twiddle of column m = lane     : tw[m*k], split likewise
   (split = ready for v_mul_hi_i32: nothing is shifted or masked per use) */
struct ConstTiles {
  const int2 *rot, *tw2, *tw3;
};

__device__ __forceinline__ int2 split_twiddle(int32_t w) {
  return make_int2((int32_t)((uint32_t)w << 16), (int32_t)((uint32_t)w & 0xffff0000u));
}

__device__ __forceinline__ void fill_const_tiles(int32_t *base, int tid, int nthreads) {
  int2 *b2 = reinterpret_cast<int2 *>(base);
  for (int i = tid; i < 512; i += nthreads) {
    int32_t X, Y;
    rot_pair((i & 63) + 64 * (i >> 6), 512, 1, X, Y);
    b2[i] = make_int2(X, Y);
  }
  for (int i = tid; i < 448; i += nthreads) {
    int k = (i >> 6) + 1, l = i & 63;
    b2[512 + i] = split_twiddle(xaac_tab_fft_tw[8 * (l & 7) * k]);
    b2[512 + 448 + i] = split_twiddle(xaac_tab_fft_tw[l * k]);
  }
}

/* twiddle with a pre-split factor */
__device__ __forceinline__ cpx twiddle2(cpx x, int2 w) {
  cpx r;
  r.re = fx_shlw(fx_sub(fx_mulhi(x.re, w.x), fx_mulhi(x.im, w.y)), 1);
  r.im = fx_shlw(fx_add(fx_mulhi(x.re, w.y), fx_mulhi(x.im, w.x)), 1);
  return r;
}

/* spec (16 words per lane, as 4 coalesced int4 rows) -> de-interleaved LDS tile:
   E[i] = spec[2i] at word i, O[i] = spec[2i+1] at word 512 + i */
__device__ __forceinline__ void stage_spec(int32_t *buf, const int4 (&v)[4], int lane) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
    *reinterpret_cast<int2 *>(buf + 128 * r + 2 * lane) = make_int2(v[r].x, v[r].z);
    *reinterpret_cast<int2 *>(buf + 512 + 128 * r + 2 * lane) = make_int2(v[r].y, v[r].w);
  }
}

/* 1024-sample transform: staged spectrum in buf -> un-windowed block y[1024] in buf */
__device__ __forceinline__ void long_transform(int32_t *buf, int lane, int e, const ConstTiles &wc) {
  cpx x[8], y[8];
  int2 *Z = reinterpret_cast<int2 *>(buf);
  const int sl = e < 0 ? -e : 0, sr = e < 0 ? 0 : e;
  /* pre-rotation of bins lane + 64k, fused with pass 1.  Lane l runs butterfly
     b = digrev8(l) so that its inputs are the conflict-free column l. */
#pragma unroll
  for (int k = 0; k < 8; k++) {
    int c = lane + 64 * k;
    int32_t a = buf[c], b = buf[512 + 511 - c];
    const int2 r = wc.rot[64 * k + lane];
    x[k].re = fx_add(fx_mulhi(a, r.x), fx_mulhi(b, r.y));
    x[k].im = fx_sub(fx_mulhi(b, r.x), fx_mulhi(a, r.y));
  }
  if (e < 0) { /* one of the two shift counts is 0 (the exponent is per frame: a scalar branch) */
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = {fx_shlw(x[k].re, sl), fx_shlw(x[k].im, sl)};
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = {x[k].re >> sr, x[k].im >> sr};
  }
  bfly8(x, y);
  {
    int A = lane & 7, B = lane >> 3;
#pragma unroll
    for (int q = 0; q < 8; q++) Z[phi(A, B, q)] = make_int2(y[q].re, y[q].im);
  }
  /* pass 2: del = 8; butterfly (group A, column m = C); column 0 is not twiddled */
  {
    int A = lane >> 3, C = lane & 7;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      int2 t = Z[phi(A, k, C)];
      x[k] = {t.x, t.y};
    }
    if (C != 0) { /* one exec-mask region instead of fourteen selects */
#pragma unroll
      for (int k = 1; k < 8; k++) x[k] = twiddle2(x[k], wc.tw2[64 * (k - 1) + lane]);
    }
    bfly8(x, y);
#pragma unroll
    for (int q = 0; q < 8; q++) Z[phi(A, q, C)] = make_int2(y[q].re, y[q].im);
  }
  /* pass 3: del = 64; butterfly column m = lane (twiddled even for m = 0: tw[0] != 1) */
  {
    int B = lane >> 3, C = lane & 7;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      int2 t = Z[phi(k, B, C)];
      x[k] = {t.x, t.y};
    }
#pragma unroll
    for (int k = 1; k < 8; k++) x[k] = twiddle2(x[k], wc.tw3[64 * (k - 1) + lane]);
    bfly8(x, y);
  }
  /* post-rotation of bins 64q + lane (same rotation registers), +-50 cross term */
  constexpr int32_t kAdjP = (int32_t)(50u << 16), kAdjN = (int32_t)((uint32_t)(uint16_t)(int16_t)-50 << 16);
#pragma unroll
  for (int q = 0; q < 8; q++) {
    int c = lane + 64 * q;
    const int2 w = wc.rot[64 * q + lane];
    int32_t r = fx_add(fx_mulhi(y[q].re, w.x), fx_mulhi(y[q].im, w.y));
    int32_t i = fx_sub(fx_mulhi(y[q].re, w.y), fx_mulhi(y[q].im, w.x));
    buf[2 * c] = fx_add(r, fx_mulhi(i, kAdjN));
    buf[1023 - 2 * c] = fx_add(i, fx_mulhi(r, kAdjP));
  }
}

/* EIGHT_SHORT: eight 128-sample transforms; lane = (window w, butterfly b) */
__device__ __forceinline__ void short_transform(lds_i32 *buf, int lane, int e) {
  cpx x[8], y[8];
  const int w = lane >> 3, b = lane & 7;
  const int sl = e < 0 ? -e : 0, sr = e < 0 ? 0 : e;
  int32_t rx[8], ry[8];
#pragma unroll
  for (int k = 0; k < 8; k++) rot_pair(b + 8 * k, 64, 8, rx[k], ry[k]);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    int c = b + 8 * k;
    int32_t a = buf[64 * w + c], bb = buf[512 + 64 * w + 63 - c];
    int32_t re = fx_add(fx_mulhi(a, rx[k]), fx_mulhi(bb, ry[k]));
    int32_t im = fx_sub(fx_mulhi(bb, rx[k]), fx_mulhi(a, ry[k]));
    x[k].re = scale_by_expo(re, sl, sr);
    x[k].im = scale_by_expo(im, sl, sr);
  }
  bfly8(x, y);
#pragma unroll
  for (int q = 0; q < 8; q++) {
    buf[2 * (64 * w + 8 * b + q)] = y[q].re;
    buf[2 * (64 * w + 8 * b + q) + 1] = y[q].im;
  }
  /* last pass (del = 8), column m = b, twiddles tw[8*m*k] */
#pragma unroll
  for (int k = 0; k < 8; k++) {
    x[k] = {buf[2 * (64 * w + b + 8 * k)], buf[2 * (64 * w + b + 8 * k) + 1]};
  }
#pragma unroll
  for (int k = 1; k < 8; k++) x[k] = twiddle(x[k], xaac_tab_fft_tw[8 * b * k]);
  bfly8(x, y);
  constexpr int32_t kAdjP = (int32_t)(402u << 16), kAdjN = (int32_t)((uint32_t)(uint16_t)(int16_t)-402 << 16);
#pragma unroll
  for (int q = 0; q < 8; q++) {
    int c = b + 8 * q;
    int32_t r = fx_add(fx_mulhi(y[q].re, rx[q]), fx_mulhi(y[q].im, ry[q]));
    int32_t i = fx_sub(fx_mulhi(y[q].re, ry[q]), fx_mulhi(y[q].im, rx[q]));
    buf[128 * w + 2 * c] = fx_add(r, fx_mulhi(i, kAdjN));
    buf[128 * w + 127 - 2 * c] = fx_add(i, fx_mulhi(r, kAdjP));
  }
}

/* ---- window / overlap-add variants (y and old overlap are in LDS) ---------- */

/* lpfuncs.c:316 */
__device__ __forceinline__ int32_t to_ovl(int32_t v, int q) { return fx_shr_rnd(v, 16 - q); }

/* block.c:1193: n-point ola1; coef -> 2n block (upper half read), prev -> n old-overlap words */
__device__ __forceinline__ void ola1(const lds_i32 *coef, const lds_i32 *prev, const Sink &sk, int obase,
                                     const lds_i16 *win, int q, int n, int lane) {
  for (int i = lane; i < n; i += 64) {
    int16_t w1 = win[2 * n - 2 * i - 1], w2 = win[2 * n - 2 * i - 2];
    int32_t c = coef[2 * n - 1 - i], p = prev[i];
    sk.put(obase + n - 1 - i, fx_sub_sat(fx_shl_dir_sat_limit(mul16(c, w2), q), nosh(p, w1)));
    sk.put(obase + n + i, fx_sub_sat(fx_shl_dir_sat_limit(mul16(fx_neg_sat(c), w1), q), nosh(p, w2)));
  }
}

/* block.c:1220: value i (0..2n-1) of the short/short overlap */
__device__ __forceinline__ int32_t ola2_value(const lds_i32 *coef, const lds_i32 *prev, const lds_i16 *win, int q,
                                              int n, int i) {
  int32_t a;
  if (i < n) {
    a = fx_sub_sat(mul16(coef[n + i], win[2 * i]), mul16(prev[n - 1 - i], win[2 * i + 1]));
  } else {
    int j = i - n;
    a = fx_sub_sat(mul16(fx_neg_sat(coef[2 * n - 1 - j]), win[2 * n - 2 * j - 1]),
                   mul16(prev[j], win[2 * n - 2 * j - 2]));
  }
  return fx_shr_rnd(a, 16 - (q + 1));
}

/* lpfuncs.c:94: long block beside a short edge (start: edge on the left) */
__device__ __forceinline__ void win_edge(const lds_i32 *y, const lds_i32 *ov, const Sink &sk, const lds_i16 *wl,
                                         const lds_i16 *ws, int q, bool start, int lane) {
  constexpr int u = 64;
  if (start) {
    for (int i = lane; i < 7 * u; i += 64) {
      int32_t t = fx_shl_dir_sat_limit(mul16(y[8 * u + i], wl[2 * i]), q + 1);
      sk.put(i, fx_add_sat(t, fx_shlw(ov[i], 16)));
      t = fx_shl_dir_sat_limit(mul16(fx_neg(y[15 * u - 1 - i]), wl[2 * (7 * u - i) - 1]), q);
      sk.put(i + 9 * u, fx_shlw(t, 1));
    }
  } else {
    for (int i = lane; i < 7 * u; i += 64) {
      sk.put(i, nosh(ov[8 * u - 1 - i], fx_neg16(wl[2 * i + 1])));
      sk.put(9 * u + i, fx_sub_sat(fx_shl_dir_sat_limit(fx_neg(y[15 * u - 1 - i]), q - 1),
                                   nosh(ov[i + u], wl[14 * u - 2 - 2 * i])));
    }
  }
  {
    const int i = lane;
    const lds_i16 *wa = start ? wl + 14 * u : ws;
    const lds_i16 *wb = start ? ws : wl + 14 * u;
    int32_t c = y[15 * u + i];
    int32_t p = start ? ov[8 * u - 1 - i] : ov[u - 1 - i];
    int16_t w1 = wa[2 * i], w2 = wa[2 * i + 1], w4 = wb[2 * i], w3 = wb[2 * i + 1];
    int32_t a = fx_sub_sat(fx_shl_dir_sat_limit(mul16(c, w1), q), nosh(p, w3));
    int32_t b = fx_sub_sat(fx_shl_dir_sat_limit(mul16(fx_neg_sat(c), w2), q), nosh(p, w4));
    sk.put(7 * u + i, fx_shlw(a, start ? 1 : 0));
    sk.put(9 * u - 1 - i, fx_shlw(b, start ? 1 : 0));
  }
}

/* lpfuncs.c:180-284: EIGHT_SHORT after a long-tailed frame; also yields new overlap[0..63] */
__device__ __forceinline__ void short_after_long(const lds_i32 *y, const lds_i32 *ov, const Sink &sk,
                                                 int32_t *ovl_out, const lds_i16 *wsc, const lds_i16 *wsp,
                                                 const lds_i16 *wlp, int q, int lane) {
  constexpr int u = 64;
  for (int i = lane; i < 7 * u; i += 64) sk.put(i, nosh(ov[8 * u - 1 - i], fx_neg16(wlp[2 * i + 1])));
  const int i = lane;
  sk.put(7 * u + i,
         fx_sub_sat(fx_shl_dir_sat_limit(mul16(y[u + i], wsp[2 * i]), q), nosh(ov[u - 1 - i], wlp[14 * u + 1 + 2 * i])));
  sk.put(8 * u + i, fx_sub_sat(fx_shl_dir_sat_limit(mul16(fx_neg_sat(y[2 * u - 1 - i]), wsp[2 * u - 2 * i - 1]), q),
                               nosh(ov[i], wlp[16 * u - 2 - 2 * i])));
  for (int b = 0; b < 4; b++) {
    int inc = 2 * u * b;
    const lds_i32 *cur = y + u + inc;
    const lds_i32 *pv = ov + u + inc;
    const lds_i16 *wl = wlp + 2 * (7 * u - inc);
    int32_t c1 = cur[2 * u + i], c2 = cur[-1 - i];
    int16_t sh1 = wsc[2 * i + 1], sh2 = wsc[2 * i];
    int32_t a = fx_sub(mul16(c1, sh2), mul16(c2, sh1));
    sk.put(9 * u + inc + i, fx_sub_sat(fx_shl_dir_sat_limit(a, q), nosh(pv[i], wl[-2 - 2 * i])));
    if (b != 3) {
      int32_t d = fx_sub(mul16(fx_neg_sat(c1), sh1), mul16(c2, sh2));
      sk.put(9 * u + inc + 2 * u - 1 - i,
             fx_sub_sat(fx_shl_dir_sat_limit(d, q), nosh(pv[2 * u - 1 - i], wl[-4 * u + 2 * i])));
    }
  }
  int32_t a = fx_sub(mul16(fx_neg(y[10 * u - 1 - i]), wsc[2 * u - 2 * i - 1]), mul16(y[6 * u + i], wsc[2 * u - 2 * i - 2]));
  ovl_out[i] = fx_round16(fx_shl_dir_sat_limit(a, q + 1));
}

/* ONLY_LONG after a short edge, LONG_START, LONG_STOP (lpfuncs.c:489-655); returns qshift_adj */
__device__ __noinline__ int long_transition_paths(const lds_i32 *y, const lds_i32 *ovs, Sink sk, int32_t *ovl,
                                                  const lds_i16 *wl, const lds_i16 *ws, int q, int seq,
                                                  bool prev_short_edge, int lane) {
  constexpr int u = 64;
  if (seq == XAAC_K_ONLY_LONG) { /* after LONG_START / EIGHT_SHORT */
    sk.qadj = 1;
    win_edge(y, ovs, sk, wl, ws, q, true, lane);
    for (int i = lane; i < 8 * u; i += 64) ovl[i] = to_ovl(y[i], q);
  } else if (seq == XAAC_K_LONG_START) {
    if (!prev_short_edge) {
      ola1(y, ovs, sk, 0, wl, q, 8 * u, lane);
    } else {
      sk.qadj = 1;
      win_edge(y, ovs, sk, wl, ws, q, true, lane);
    }
    for (int i = lane; i < 7 * u; i += 64) ovl[i] = fx_shr_rnd(fx_neg_sat(y[8 * u - 1 - i]), 16 - q);
    ovl[7 * u + lane] = to_ovl(y[lane], q);
  } else { /* LONG_STOP */
    if (prev_short_edge) {
      for (int i = lane; i < 7 * u; i += 64) sk.put(i, fx_shl_sat((int16_t)ovs[i], 15));
      ola1(y + 14 * u, ovs + 7 * u, sk, 7 * u, ws, q, u, lane);
      for (int i = lane; i < 7 * u; i += 64)
        sk.put(9 * u + i, fx_shl_dir_sat_limit(fx_neg_sat(y[15 * u - 1 - i]), q - 1));
    } else {
      win_edge(y, ovs, sk, wl, ws, q, false, lane);
    }
    for (int i = lane; i < 8 * u; i += 64) ovl[i] = to_ovl(y[i], q);
  }
  return sk.qadj;
}

/* EIGHT_SHORT (lpfuncs.c:657-798): transform + all its overlap handling; qshift_adj is always 2 */
__device__ __noinline__ void eight_short_path(lds_i32 *buf, const lds_i32 *ovs, Sink sk, int32_t *ovl,
                                              const lds_i16 *wl, const lds_i16 *ws, const lds_i16 *wsc,
                                              int headroom, bool prev_short_edge, int lane) {
  constexpr int u = 64;
  const int e = 5 - (headroom - 1);
  const int q = e + 2 + 8;
  short_transform(buf, lane, e);
  const lds_i32 *y = buf;
  const int i = lane;
  if (prev_short_edge) {
    for (int n = lane; n < 7 * u; n += 64) sk.put(n, fx_shl_sat((int16_t)ovs[n], 15));
    ola1(y, ovs + 7 * u, sk, 7 * u, ws, q, u, lane);
    for (int b = 0; b < 3; b++) {
      /* ola1 against the (requantised) tail of the previous short window */
      const lds_i32 *coef = y + 2 * u + 2 * u * b;
      int16_t w1 = wsc[2 * u - 2 * i - 1], w2 = wsc[2 * u - 2 * i - 2];
      int32_t c = coef[2 * u - 1 - i], pr = to_ovl(y[2 * u * b + i], q);
      sk.put(9 * u + 2 * u * b + u - 1 - i, fx_sub_sat(fx_shl_dir_sat_limit(mul16(c, w2), q), nosh(pr, w1)));
      sk.put(9 * u + 2 * u * b + u + i, fx_sub_sat(fx_shl_dir_sat_limit(mul16(fx_neg_sat(c), w1), q), nosh(pr, w2)));
    }
    int32_t t_lo = ola2_value(y + 8 * u, y + 6 * u, wsc, q, u, i);
    int32_t t_hi = ola2_value(y + 8 * u, y + 6 * u, wsc, q, u, u + i);
    sk.put(15 * u + i, fx_shl_sat((int16_t)t_lo, 15)); /* lpfuncs.c:335 */
    ovl[i] = t_hi;
  } else {
    short_after_long(y, ovs, sk, ovl, wsc, ws, wl, q, lane);
  }
  for (int b = 0; b < 3; b++) {
    ovl[u + 2 * u * b + i] = ola2_value(y + 10 * u + 2 * u * b, y + 8 * u + 2 * u * b, wsc, q, u, i);
    ovl[u + 2 * u * b + u + i] = ola2_value(y + 10 * u + 2 * u * b, y + 8 * u + 2 * u * b, wsc, q, u, u + i);
  }
  ovl[7 * u + i] = to_ovl(y[14 * u + i], q);
}

}  // namespace

/* ========================================================================= */
/* hot-path OLA (aac_imdct.c:506) for one group of four t's.  QPOS = (q_shift > 0)
   selects the reference's two branches at compile time. */
/* FAST: the wave has checked that no windowed value saturates in the left shift and that every old-overlap word
   is within +-65535 (so overlap x window fits 32 bits: one v_mul_i32_i24 instead of a clamped 64-bit product) */
template <bool QPOS, bool FAST>
__device__ __forceinline__ void ola_long_long4(const int32_t *y, const int16_t *wl, int t0, const int4 &old4, int q,
                                               int32_t (&lo)[4], int32_t (&hi)[4], int4 &new4) {
  const int4 vv = *reinterpret_cast<const int4 *>(y + 1020 - t0);       /* y[1023-t], j = 3..0 */
  const int4 uu = *reinterpret_cast<const int4 *>(y + t0);              /* y[t]                */
  const int4 ww = *reinterpret_cast<const int4 *>(wl + 2 * (508 - t0)); /* W32[511-t], j = 3..0 */
  const int32_t vj[4] = {vv.w, vv.z, vv.y, vv.x};
  const int32_t wj[4] = {ww.w, ww.z, ww.y, ww.x};
  const int32_t oj[4] = {old4.x, old4.y, old4.z, old4.w};
  const int32_t uj[4] = {uu.x, uu.y, uu.z, uu.w};
  int32_t nv[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    int32_t a = fx_mul32xlo(vj[j], wj[j]);
    int32_t b = fx_mul32xhi(FAST ? fx_neg(vj[j]) : fx_neg_sat(vj[j]), wj[j]); /* FAST: |y| is far from MIN */
    int32_t o = oj[j];
    if (QPOS) {
      a = FAST ? fx_shlw(a, q) : fx_shl_sat(a, q);
      b = FAST ? fx_shlw(b, q) : fx_shl_sat(b, q);
    } else {
      a = fx_shr(a, -q);
      b = fx_shr(b, -q);
      o = (int16_t)o; /* aac_imdct.c:679: this branch reads the overlap word as WORD16 */
    }
    if (FAST || !QPOS) {
      lo[j] = fx_sub_sat(a, __mul24(o, (int16_t)(wj[j] >> 16))); /* out[511-t] */
      hi[j] = fx_sub_sat(b, __mul24(o, (int16_t)wj[j]));         /* out[512+t] */
    } else {
      lo[j] = fx_sub_sat(a, nosh(o, (int16_t)(wj[j] >> 16)));
      hi[j] = fx_sub_sat(b, nosh(o, (int16_t)wj[j]));
    }
    nv[j] = to_ovl(uj[j], q);
  }
  new4 = make_int4(nv[0], nv[1], nv[2], nv[3]);
}

__device__ __forceinline__ int32_t pack16(int16_t a, int16_t b) {
  return (int32_t)((uint32_t)(uint16_t)a | ((uint32_t)(uint16_t)b << 16));
}

/* One work unit = one access unit = CF channel-frames handled back to back by
   the same wave, so that for CF = 2 the PCM of both channels leaves as
   interleaved 16-byte stores (channel 0 is parked in LDS meanwhile).
   MODE = XAAC_PCM_LC / XAAC_PCM_SBR. */
template <int CF, int MODE>
__global__ __launch_bounds__(XAAC_IMDCT_BLOCK, XAAC_IMDCT_MIN_WAVES_PER_SIMD) void xaac_imdct_ola_kernel(
    XaacImdctParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int16_t *s_win = reinterpret_cast<int16_t *>(smem); /* [long sine|long kbd|short sine|short kbd] */
  const int tid = threadIdx.x, lane_id = tid & 63;
  /* wave index as a scalar: everything per-frame below (pointers, window sequence,
     block exponent, branches) then lives in SGPRs and branches are s_cbranch, not exec masks */
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int32_t *s_const = reinterpret_cast<int32_t *>(smem + XAAC_IMDCT_LDS_WIN_BYTES);
  int32_t *buf = s_const + XAAC_IMDCT_LDS_CONST_WORDS + wave * XAAC_IMDCT_LDS_WAVE_WORDS;
  int32_t *ovs = buf + 1024; /* 2 KB: old-overlap copy for the rare paths / parked channel-0 PCM */
  int16_t *park = reinterpret_cast<int16_t *>(ovs);

  for (int i = tid; i < 1024; i += XAAC_IMDCT_BLOCK) {
    s_win[i] = xaac_tab_win_long_sine[i];
    s_win[1024 + i] = xaac_tab_win_long_kbd[i];
  }
  for (int i = tid; i < 128; i += XAAC_IMDCT_BLOCK) {
    s_win[2048 + i] = xaac_tab_win_short_sine[i];
    s_win[2176 + i] = xaac_tab_win_short_kbd[i];
  }
  fill_const_tiles(s_const, tid, XAAC_IMDCT_BLOCK);
  const ConstTiles wc = {reinterpret_cast<const int2 *>(s_const), reinterpret_cast<const int2 *>(s_const) + 512,
                         reinterpret_cast<const int2 *>(s_const) + 512 + 448};
  __syncthreads();

  const int waves_total = gridDim.x * XAAC_IMDCT_WAVES;
  const int n_au = p.n_ch / CF;
  for (int au = blockIdx.x * XAAC_IMDCT_WAVES + wave; au < n_au; au += waves_total) {
    bool parked = false; /* CF == 2: channel 0's PCM sits in LDS waiting for channel 1 */
#pragma unroll 1
    for (int c = 0; c < CF; c++) {
      const int ch = au * CF + c;
      /* keep lane-derived LDS addresses out of loop-invariant registers: recomputing the
         ~30 of them per frame is a few dozen VALU ops, holding them costs a wave of occupancy */
      int lane = lane_id;
      asm volatile("" : "+v"(lane));
      /* ---- loads: 4 KB spectrum + 2 KB overlap, 16 B per lane per instruction */
      const int4 *sp = reinterpret_cast<const int4 *>(p.spec + (size_t)ch * 1024);
      int32_t *ovl = p.overlap + (size_t)ch * 512;
      int4 v[4], o4[2];
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = sp[64 * r + lane];
      const int ics_bits = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const uint16_t *>(p.ics + ch));
      const int st_bits = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const uint16_t *>(p.state + ch));
      const int seq = ics_bits & 0xff, shape = ics_bits >> 8;
      const int pseq = st_bits & 0xff, pshape = st_bits >> 8;
      /* window_sequence is a 2-bit, window_shape a 1-bit field of the bitstream (ixheaacd_channel.c:ics_info); values
         beyond that cannot come from a parser and would index past the window tables: the channel-frame is refused
         -- overlap, state and output untouched, XAAC_FATAL_BAD_WINDOW_SEQ in its status word -- and its neighbours
         are not affected (scalar test: both words are wave-uniform) */
      if (((ics_bits | st_bits) & ~0x0103) != 0) {
        if (CF == 2 && parked) { /* channel 0 of this access unit is waiting in LDS: let it out alone */
          for (int n = lane; n < 1024; n += 64) p.pcm16[(size_t)au * 2048 + 2 * (size_t)n] = park[n];
          parked = false;
        }
        if (lane == 0 && p.status) p.status[ch] = XAAC_FATAL_BAD_WINDOW_SEQ;
        continue;
      }

      /* block exponent: norm32 of the OR of abs_nrm over the frame (aac_tns.c:422).  Only the OR's top bit
         matters, and that is the top bit of max abs_nrm(x) = max(max x, ~min x) */
      int32_t mx = v[0].x, mn = v[0].x;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        mx = max(max(mx, v[r].x), max(v[r].y, max(v[r].z, v[r].w)));
        mn = min(min(mn, v[r].x), min(v[r].y, min(v[r].z, v[r].w)));
      }
      const int headroom = fx_norm32(wave_max_nonneg(max(mx, ~mn)));

      stage_spec(buf, v, lane);
      /* old overlap: issued now, consumed after the transform (its latency hides under the FFT) */
      o4[0] = reinterpret_cast<const int4 *>(ovl)[lane];
      o4[1] = reinterpret_cast<const int4 *>(ovl)[64 + lane];

      const bool prev_short_edge = (pseq == XAAC_K_LONG_START) || (pseq == XAAC_K_EIGHT_SHORT);
      const bool hot = (seq == XAAC_K_ONLY_LONG) && !prev_short_edge;
      const int16_t *wl = s_win + 1024 * pshape;       /* previous shape, long */
      const int16_t *ws = s_win + 2048 + 128 * pshape; /* previous shape, short */
      const size_t obase = (size_t)au * 1024 * CF + c;
      Sink sk;
      sk.o32 = p.out32 ? p.out32 + obase : nullptr;
      sk.p16 = p.pcm16 ? p.pcm16 + obase : nullptr;
      sk.stride = CF;
      sk.mode = MODE;
      sk.qadj = 2;
      sk.peer = nullptr;

      /* CF == 2: channel 0 was parked but this channel cannot pair with it (rare path,
         which also needs the parking area for the old overlap): flush channel 0 now */
      if (CF == 2 && parked && !hot) {
        for (int n = lane; n < 1024; n += 64) p.pcm16[obase - 1 + 2 * (size_t)n] = park[n];
        parked = false;
      }
      /* stereo + SBR hand-off, channel 1: the in-place conversion's view of channel 0 (see Sink).  Channel 0's PCM is in
         LDS while it is parked, else in the output buffer, where this wave's own stores have to have landed first. */
      const bool in_place1 = CF == 2 && MODE == XAAC_PCM_SBR && c == 1 && p.pcm16 != nullptr;
      if (in_place1 && !parked) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sk.peer = p.pcm16 + (size_t)au * 2048;
      }

      if (seq != XAAC_K_EIGHT_SHORT) {
        const int e = 8 - (headroom - 1);
        const int q = e + 2 + 5;
        long_transform(buf, lane, e, wc);
        const int32_t *y = buf;

        if (hot) {
          /* ---- hot path: lane owns t = 4l+j and 256+4l+j, i.e. outputs 508-t0.. and 512+t0.. */
#pragma unroll
          for (int g = 0; g < 2; g++) {
            const int t0 = 256 * g + 4 * lane;
            int32_t lo[4], hi[4];
            int4 nv;
            if (q > 0) {
              /* |y * w >> 16| <= |y| / 2 + 1: with the block's largest |y[512..1023]| the shift cannot saturate when
                 (max >> 1) + 1 <= MAX >> q; the old overlap must fit +-65535 for the 24-bit product */
              const int4 vv = *reinterpret_cast<const int4 *>(y + 1020 - t0);
              const int32_t ymx = max(max(vv.x, vv.y), max(vv.z, vv.w)), ymn = min(min(vv.x, vv.y), min(vv.z, vv.w));
              const int32_t omx = max(max(o4[g].x, o4[g].y), max(o4[g].z, o4[g].w));
              const int32_t omn = min(min(o4[g].x, o4[g].y), min(o4[g].z, o4[g].w));
              const int32_t ylim = ((FX_MAX32 >> q) - 1) << 1;
              const bool slow = ymx > ylim || ymn < -ylim || omx > 65535 || omn < -65535;
              if (!__ballot(slow))
                ola_long_long4<true, true>(y, wl, t0, o4[g], q, lo, hi, nv);
              else
                ola_long_long4<true, false>(y, wl, t0, o4[g], q, lo, hi, nv);
            } else {
              ola_long_long4<false, false>(y, wl, t0, o4[g], q, lo, hi, nv);
            }
            reinterpret_cast<int4 *>(ovl)[64 * g + lane] = nv;
            if (sk.o32) {
              if (CF == 1) {
                *reinterpret_cast<int4 *>(sk.o32 + 508 - t0) = make_int4(lo[3], lo[2], lo[1], lo[0]);
                *reinterpret_cast<int4 *>(sk.o32 + 512 + t0) = make_int4(hi[0], hi[1], hi[2], hi[3]);
              } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                  sk.o32[CF * (511 - t0 - j)] = lo[j];
                  sk.o32[CF * (512 + t0 + j)] = hi[j];
                }
              }
            }
            if (sk.p16) {
              int16_t pl[4], ph[4];
#pragma unroll
              for (int j = 0; j < 4; j++) {
                int32_t lv = lo[j]; /* sample 511 - t0 - j: below 512, where channel 1 sees channel 0's halfwords */
                if (in_place1) {
                  const int m = 2 * (511 - t0 - j) + 1;
                  const int16_t c0 = parked ? park[m] : sk.peer[2 * m];
                  lv = (int32_t)(((uint32_t)lv & 0xffff0000u) | (uint16_t)c0);
                }
                pl[j] = sk.to_pcm(lv);
                ph[j] = sk.to_pcm(hi[j]);
              }
              if (CF == 1) {
                *reinterpret_cast<int2 *>(sk.p16 + 508 - t0) = make_int2(pack16(pl[3], pl[2]), pack16(pl[1], pl[0]));
                *reinterpret_cast<int2 *>(sk.p16 + 512 + t0) = make_int2(pack16(ph[0], ph[1]), pack16(ph[2], ph[3]));
              } else if (c == 0) {
                /* park: channel 1 (same lane, same samples) will interleave and store */
                *reinterpret_cast<int2 *>(park + 508 - t0) = make_int2(pack16(pl[3], pl[2]), pack16(pl[1], pl[0]));
                *reinterpret_cast<int2 *>(park + 512 + t0) = make_int2(pack16(ph[0], ph[1]), pack16(ph[2], ph[3]));
              } else if (parked) {
                const int2 a = *reinterpret_cast<const int2 *>(park + 508 - t0);
                const int2 b = *reinterpret_cast<const int2 *>(park + 512 + t0);
                const int16_t l0[4] = {(int16_t)a.x, (int16_t)(a.x >> 16), (int16_t)a.y, (int16_t)(a.y >> 16)};
                const int16_t l1[4] = {(int16_t)b.x, (int16_t)(b.x >> 16), (int16_t)b.y, (int16_t)(b.y >> 16)};
                int32_t *dst = reinterpret_cast<int32_t *>(p.pcm16 + (size_t)au * 2048); /* one dword per sample pair */
                *reinterpret_cast<int4 *>(dst + 508 - t0) =
                    make_int4(pack16(l0[0], pl[3]), pack16(l0[1], pl[2]), pack16(l0[2], pl[1]), pack16(l0[3], pl[0]));
                *reinterpret_cast<int4 *>(dst + 512 + t0) =
                    make_int4(pack16(l1[0], ph[0]), pack16(l1[1], ph[1]), pack16(l1[2], ph[2]), pack16(l1[3], ph[3]));
              } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                  sk.p16[CF * (511 - t0 - j)] = pl[j];
                  sk.p16[CF * (512 + t0 + j)] = ph[j];
                }
              }
            }
          }
          parked = (CF == 2 && c == 0 && sk.p16 != nullptr);
        } else {
          /* rare long-window transitions: out of line so that their register
             appetite does not set the occupancy of the hot path */
#ifndef XAAC_HOT_ONLY /* (analysis builds drop the rare paths to read the hot loop's ISA) */
          reinterpret_cast<int4 *>(ovs)[lane] = o4[0];
          reinterpret_cast<int4 *>(ovs)[64 + lane] = o4[1];
          sk.qadj = long_transition_paths((const lds_i32 *)buf, (const lds_i32 *)ovs, sk, ovl, (const lds_i16 *)wl,
                                          (const lds_i16 *)ws, q, seq, prev_short_edge, lane);
#endif
        }
      } else {
#ifndef XAAC_HOT_ONLY
        reinterpret_cast<int4 *>(ovs)[lane] = o4[0];
        reinterpret_cast<int4 *>(ovs)[64 + lane] = o4[1];
        eight_short_path((lds_i32 *)buf, (const lds_i32 *)ovs, sk, ovl, (const lds_i16 *)wl, (const lds_i16 *)ws,
                         (const lds_i16 *)(s_win + 2048 + 128 * shape), headroom, prev_short_edge, lane);
#endif
      }

      if (lane == 0) {
        p.state[ch].window_sequence = (uint8_t)seq;
        p.state[ch].window_shape = (uint8_t)shape;
        if (p.qshift_adj) p.qshift_adj[ch] = (int8_t)sk.qadj;
        if (p.status) p.status[ch] = XAAC_OK;
      }
    }
  }
}

namespace {
typedef void (*imdct_kernel_t)(XaacImdctParams);
imdct_kernel_t pick_kernel(int ch_fac, int pcm_mode) {
  if (ch_fac == 2) return pcm_mode ? xaac_imdct_ola_kernel<2, 1> : xaac_imdct_ola_kernel<2, 0>;
  return pcm_mode ? xaac_imdct_ola_kernel<1, 1> : xaac_imdct_ola_kernel<1, 0>;
}
}  // namespace

extern "C" hipError_t xaac_launch_imdct(const XaacImdctParams *p, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(pick_kernel(p->ch_fac, p->pcm_mode), dim3(grid), dim3(XAAC_IMDCT_BLOCK), XAAC_IMDCT_LDS_BYTES,
                     stream, *p);
  return hipGetLastError();
}

/* resident workgroups per CU for this kernel (registers + LDS), for sizing the persistent grid */
extern "C" int xaac_imdct_blocks_per_cu(void) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, xaac_imdct_ola_kernel<2, 0>, XAAC_IMDCT_BLOCK,
                                                   XAAC_IMDCT_LDS_BYTES) != hipSuccess || n < 1)
    n = 2;
  return n;
}

/* xaac_warm_up (xaac_abi.cpp): asking for a kernel's attributes puts this translation unit's code object on the device */
extern "C" hipError_t xaac_warm_imdct(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(pick_kernel(1, XAAC_PCM_SBR)));
}
