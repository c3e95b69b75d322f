/*
 * usac_imdct.h -- the USAC frequency-domain IMDCT (ccfl = 1024, no FAC, previous frame FD), shared by the gfx950 kernel
 * (usac_imdct_kernel.hip) and, compiled for the host, by the checker (oracle/oracle_usac.cpp).
 *
 * Restates, as per-index functions (every butterfly of a pass and every output sample is independent, so a caller may
 * run them in any order or all at once):
 *   ixheaacd_fd_imdct_long / _short       decoder/ixheaacd_imdct.c:477 / :336
 *   ixheaacd_acelp_imdct, _fft_based_imdct, pre / post twiddle      ixheaacd_imdct.c:186 / :149 / :111 / :129
 *   ixheaacd_complex_fft_p2_dec (fft_mode = 1 branch)               ixheaacd_fft.c:1412, :1966-2484
 *   ixheaacd_windowing_long1 / _long3 / _short2 / _short3 / _short4, _scale_down(_adj)   ixheaacd_basic_ops.c:77-660
 * Arithmetic: saturating adds / subtractions and (a*b)>>31 products clamped to 32 bits inside the FFT, truncating
 * (a*b)>>32 products in the twiddles, (a*b)>>31 wrapped to 32 bits in the windows -- bit for bit the reference's.
 */
#ifndef XAAC_USAC_IMDCT_H
#define XAAC_USAC_IMDCT_H

#include "fx.h"

#ifndef XAAC_USAC_TABLES_INCLUDED
#define XAAC_USAC_TABLES_INCLUDED
#if defined(__HIPCC__)
#define XAAC_TAB_QUAL static __device__ const
#include "tables_usac.inc"
#undef XAAC_TAB_QUAL
#else
#include "tables_usac.inc"
#endif
#endif

#define XU_SHIFT_OLAP 14

/* ixheaacd_fft.c:48-56: (a*b) >> 31 clamped */
FX_HD int32_t xu_mul_sat(int32_t a, int32_t b) { return fx_sat64(((int64_t)a * (int64_t)b) >> 31); }
/* ixheaacd_basic_ops.c:28-36: (a*b) >> 31 wrapped */
FX_HD int32_t xu_mul_sh1(int32_t a, int32_t b) { return (int32_t)(((int64_t)a * (int64_t)b) >> 31); }
FX_HD int32_t xu_shl1(int32_t a) { return fx_shl_sat(a, 1); }
/* C's truncating division by a power of two (fft.c:1443-1446, :2438-2443) */
FX_HD int32_t xu_div_pow2(int32_t a, int s) { return (int32_t)((a + ((a >> 31) & ((1 << s) - 1))) >> s); }

struct XuCx {
  int32_t r, i;
};

/* the three twiddle forms of the inverse transform; h = tw[2 idx] (-sin), l = tw[2 idx + 1] (cos) */
FX_HD XuCx xu_rot0(XuCx x, int32_t h, int32_t l) { /* fft.c:2117-2120 */
  XuCx y;
  y.r = fx_add_sat(xu_mul_sat(x.r, l), xu_mul_sat(x.i, h));
  y.i = fx_add_sat(fx_neg(xu_mul_sat(x.r, h)), xu_mul_sat(x.i, l));
  return y;
}
FX_HD XuCx xu_rot1(XuCx x, int32_t h, int32_t l) { /* fft.c:2208-2212 */
  XuCx y;
  y.r = fx_sub_sat(xu_mul_sat(x.r, h), xu_mul_sat(x.i, l));
  y.i = fx_add_sat(xu_mul_sat(x.r, l), xu_mul_sat(x.i, h));
  return y;
}
FX_HD XuCx xu_rot2(XuCx x, int32_t h, int32_t l) { /* fft.c:2374-2377 */
  XuCx y;
  y.r = fx_neg(fx_add_sat(xu_mul_sat(x.r, l), xu_mul_sat(x.i, h)));
  y.i = fx_add_sat(fx_neg(xu_mul_sat(x.r, h)), xu_mul_sat(x.i, l));
  return y;
}

/* the radix-4 butterfly of every pass (fft.c:1999-2016); alt: the form of the last twiddle quadrant (:2385-2388).
   Results in the reference's store order: slot 0 <- x0, 1 <- x2, 2 <- x1, 3 <- (x3i, x3r). */
FX_HD void xu_bfly4(XuCx &a, XuCx &b, XuCx &c, XuCx &d, bool alt) {
  int32_t x0r = a.r, x0i = a.i, x1r = b.r, x1i = b.i, x2r = c.r, x2i = c.i, x3r = d.r, x3i = d.i;
  x0r = fx_add_sat(x0r, x2r);
  x0i = fx_add_sat(x0i, x2i);
  x2r = fx_sub_sat(x0r, xu_shl1(x2r));
  x2i = fx_sub_sat(x0i, xu_shl1(x2i));
  x1r = fx_add_sat(x1r, x3r);
  if (!alt) {
    x1i = fx_add_sat(x1i, x3i);
    x3r = fx_sub_sat(x1r, xu_shl1(x3r));
    x3i = fx_sub_sat(x1i, xu_shl1(x3i));
  } else {
    x1i = fx_sub_sat(x1i, x3i);
    x3r = fx_sub_sat(x1r, xu_shl1(x3r));
    x3i = fx_add_sat(x1i, xu_shl1(x3i));
  }
  x0r = fx_add_sat(x0r, x1r);
  x0i = fx_add_sat(x0i, x1i);
  x1r = fx_sub_sat(x0r, xu_shl1(x1r));
  x1i = fx_sub_sat(x0i, xu_shl1(x1i));
  x2r = fx_sub_sat(x2r, x3i);
  x2i = fx_add_sat(x2i, x3r);
  x3i = fx_add_sat(x2r, xu_shl1(x3i));
  x3r = fx_sub_sat(x2i, xu_shl1(x3r));
  a.r = x0r; a.i = x0i;
  b.r = x2r; b.i = x2i;
  c.r = x1r; c.i = x1i;
  d.r = x3i; d.i = x3r;
}

/* base-4 digit reversal of the reference's DIG_REV (fft.c:38-46) */
FX_HD unsigned xu_dig_rev(unsigned i, int m) {
  unsigned v = i;
  v = ((v & 0x33333333u) << 2) | ((v & ~0x33333333u) >> 2);
  v = ((v & 0x0F0F0F0Fu) << 4) | ((v & ~0x0F0F0F0Fu) >> 4);
  v = ((v & 0x00FF00FFu) << 8) | ((v & ~0x00FF00FFu) >> 8);
  return v >> m;
}

/* N = 512 or 64 complex points.  The transform of one block is:
 *   xu_fft_first<N>(x, y, b)      b = 0 .. N/4-1     x: N interleaved (re, im) words, already divided; y: the same size
 *   for del = 4, 16, (64):  xu_fft_pass<N>(y, del, b)   b = 0 .. N/4-1
 *   N == 512:               xu_fft_last512(y, b)        b = 0 .. 255
 * with a barrier between passes.  Input division: xu_fft_in_shift<N>(); the exponent the reference reports:
 * xu_fft_out_shift<N>(). */
template <int N> FX_HD constexpr int xu_fft_in_shift() { return N == 512 ? 6 : 5; }   /* fft.c:1438-1441 */
template <int N> FX_HD constexpr int xu_fft_out_shift() { return N == 512 ? 7 : 5; }  /* + 1 for the radix-2 pass, :2410 */

template <int N, class Mem>
FX_HD void xu_fft_first(const Mem &x, const Mem &y, int b) {
  constexpr int rev_shift = N == 512 ? 6 : 9; /* norm32(N) + 1 - 16 */
  unsigned h2 = xu_dig_rev((unsigned)(4 * b), rev_shift);
  if (N == 512) h2 = (h2 + 1) & ~1u;
  XuCx v[4];
  for (int q = 0; q < 4; q++) {
    v[q].r = x[h2 + q * (N / 2)];
    v[q].i = x[h2 + q * (N / 2) + 1];
  }
  xu_bfly4(v[0], v[1], v[2], v[3], false);
  for (int q = 0; q < 4; q++) {
    y[8 * b + 2 * q] = v[q].r;
    y[8 * b + 2 * q + 1] = v[q].i;
  }
}

template <int N, class Mem>
FX_HD void xu_fft_pass(const Mem &y, int del, int b) {
  const int jj = b % del, k = b / del;       /* twiddle column, group */
  const int p0 = 4 * del * k + jj;           /* complex index of the first leg; the others del apart */
  XuCx v[4];
  for (int q = 0; q < 4; q++) {
    v[q].r = y[2 * (p0 + q * del)];
    v[q].i = y[2 * (p0 + q * del) + 1];
  }
  bool alt = false;
  if (jj) {
    const int j = jj * (256 / del);          /* nodespacing * jj; nodespacing * del = 256 in every pass */
    const int32_t *tw = xaac_usac_fft_tw;
    v[1] = xu_rot0(v[1], tw[2 * j], tw[2 * j + 1]);
    if (j <= 128) v[2] = xu_rot0(v[2], tw[4 * j], tw[4 * j + 1]);
    else v[2] = xu_rot1(v[2], tw[4 * j - 512], tw[4 * j - 511]);
    if (j <= 85) v[3] = xu_rot0(v[3], tw[6 * j], tw[6 * j + 1]);
    else if (j <= 170) v[3] = xu_rot1(v[3], tw[6 * j - 512], tw[6 * j - 511]);
    else {
      v[3] = xu_rot2(v[3], tw[6 * j - 1024], tw[6 * j - 1023]);
      alt = true;
    }
  }
  xu_bfly4(v[0], v[1], v[2], v[3], alt);
  for (int q = 0; q < 4; q++) {
    y[2 * (p0 + q * del)] = v[q].r;
    y[2 * (p0 + q * del) + 1] = v[q].i;
  }
}

/* the radix-2 pass of the 512-point transform (fft.c:2407-2470) */
template <class Mem>
FX_HD void xu_fft_last512(const Mem &y, int b) {
  const int j = b & 127, p = b;
  const int32_t h = xaac_usac_fft_tw[4 * j], l = xaac_usac_fft_tw[4 * j + 1];
  XuCx a = {y[2 * p], y[2 * p + 1]}, c = {y[2 * (p + 256)], y[2 * (p + 256) + 1]};
  c = b < 128 ? xu_rot0(c, h, l) : xu_rot1(c, h, l);
  y[2 * (p + 256)] = fx_sub(xu_div_pow2(a.r, 1), xu_div_pow2(c.r, 1));
  y[2 * (p + 256) + 1] = fx_sub(xu_div_pow2(a.i, 1), xu_div_pow2(c.i, 1));
  y[2 * p] = fx_add(xu_div_pow2(a.r, 1), xu_div_pow2(c.r, 1));
  y[2 * p + 1] = fx_add(xu_div_pow2(a.i, 1), xu_div_pow2(c.i, 1));
}

/* pre twiddle of line pair i of an N-point block (imdct.c:111-127): x = the 2N spectral lines of the block; writes the
   FFT's divided input */
template <int N>
FX_HD XuCx xu_pre_twiddle(int32_t xa /* x[2i] */, int32_t xb /* x[2N-1-2i] */, int i) {
  const int32_t c = (N == 512 ? xaac_usac_pre_cos_512 : xaac_usac_pre_cos_64)[i];
  const int32_t s = (N == 512 ? xaac_usac_pre_sin_512 : xaac_usac_pre_sin_64)[i];
  XuCx y;
  y.r = fx_sub(fx_mulhi(fx_neg_sat(xa), c), fx_mulhi(xb, s));
  y.i = fx_sub(fx_mulhi(xb, c), fx_mulhi(xa, s));
  y.r = xu_div_pow2(y.r, xu_fft_in_shift<N>());
  y.i = xu_div_pow2(y.i, xu_fft_in_shift<N>());
  return y;
}
/* post twiddle (imdct.c:129-147): -> the new x[2i] (.r) and x[2N-1-2i] (.i) */
template <int N>
FX_HD XuCx xu_post_twiddle(XuCx v, int i) {
  const int32_t c = (N == 512 ? xaac_usac_pre_cos_512 : xaac_usac_pre_cos_64)[i];
  const int32_t s = (N == 512 ? xaac_usac_pre_sin_512 : xaac_usac_pre_sin_64)[i];
  XuCx y;
  y.r = fx_neg(fx_sub(fx_mulhi(v.r, c), fx_mulhi(v.i, s)));
  y.i = fx_neg(fx_add(fx_mulhi(v.i, c), fx_mulhi(v.r, s)));
  return y;
}

/* exponent after ixheaacd_acelp_imdct (imdct.c:186-208): *qshift -= (shift_out - log2(N2)) + 2, N2 = lines per block */
template <int N> FX_HD constexpr int xu_imdct_q_gain() { return (N == 512 ? 10 : 7) - xu_fft_out_shift<N>() - 2; }

/* ixheaacd_normalize with the count the second renormalisation can reach (imdct.c:94-100, :517): max_shift - 1 is -1
   when the transform's peak already fills the word -- a negative shift count, undefined in C; here: count & 31 */
FX_HD int32_t xu_normalize(int32_t v, int shift) { return fx_shlw(v, shift & 31); }

FX_HD const int32_t *xu_window(int len, int shape) { /* ixheaacd_calc_window for the two lengths of ccfl 1024 */
  if (len == 1024) return shape ? xaac_usac_kbd_win_1024 : xaac_usac_sine_win_1024;
  return shape ? xaac_usac_kbd_win_128 : xaac_usac_sine_win_128;
}

/* ---- long blocks: output sample i (0..1023) of the frame, before the final rescale ----------------------------------
 * x: the 1024 transform outputs after the second renormalisation; ov: the overlap (Q14); shiftp: their exponent.
 * ONLY_LONG / LONG_START: windowing_long1 (basic_ops.c:77); LONG_STOP / STOP_START: windowing_long3 (:298), no FAC. */
template <class Mem, class Ov>
FX_HD int32_t xu_long_sample(const Mem &x, const Ov &ov, int i, int shiftp, bool stop_like, int shape_prev) {
  const int d = shiftp - XU_SHIFT_OLAP; /* > 0: the transform side is shifted down; <= 0: the overlap side */
  if (!stop_like) {
    const int32_t *w = xu_window(1024, shape_prev);
    const int m = i < 512 ? i : 1023 - i;             /* the loop index of basic_ops.c:85 */
    const int32_t src1 = x[512 + m];
    const int32_t t = i < 512 ? xu_mul_sh1(src1, w[m]) : xu_mul_sh1(fx_neg_sat(src1), w[1023 - m]);
    const int32_t o = i < 512 ? xu_mul_sh1(ov[m], w[1023 - m]) : xu_mul_sh1(ov[1023 - m], w[m]);
    return d > 0 ? fx_add_sat(t >> d, o) : fx_add_sat(t, o >> -d);
  }
  const int32_t *w = xu_window(128, shape_prev);
  if (i < 448) return d > 0 ? ov[i] : (ov[i] >> -d);
  if (i < 576) {
    const int32_t src = i < 512 ? x[512 + i] : fx_neg_sat(x[512 + 1023 - i]);
    const int32_t t = xu_mul_sh1(src, w[i - 448]), o = xu_mul_sh1(ov[i], w[127 - (i - 448)]);
    return d > 0 ? fx_add_sat(t >> d, o) : fx_add_sat(t, o >> -d);
  }
  const int32_t v = fx_neg_sat(x[512 + 1023 - i]);
  return d > 0 ? (v >> d) : v;
}
FX_HD int xu_long_output_q(int shiftp) { return shiftp > XU_SHIFT_OLAP ? XU_SHIFT_OLAP : shiftp; }
/* the new overlap, sample i (imdct.c:563-576: both branches shift right) */
template <class Mem>
FX_HD int32_t xu_long_overlap(const Mem &x, int i, int shiftp) {
  const int d = shiftp > XU_SHIFT_OLAP ? shiftp - XU_SHIFT_OLAP : XU_SHIFT_OLAP - shiftp;
  const int m = i >= 512 ? i - 512 : 511 - i;
  return fx_neg_sat(x[m]) >> d;
}
/* ixheaacd_scale_down_adj(.., output_q, 15) (basic_ops.c:640): the frame's Q15 output */
FX_HD int32_t xu_scale_adj(int32_t v, int output_q) {
  return fx_add_sat(output_q > 15 ? (v >> (output_q - 15)) : fx_shl_sat(v, 15 - output_q), 11);
}
/* ixheaacd_scale_down (basic_ops.c:622) */
FX_HD int32_t xu_scale(int32_t v, int from_q, int to_q) { return from_q > to_q ? (v >> (from_q - to_q)) : fx_shl_sat(v, to_q - from_q); }

/* ---- EIGHT_SHORT frames: sample p (0..2047) of the reference's overlap_data_buf after the eight windowed blocks ------
 * x: the 8 x 128 transform outputs after the renormalisation, ov: the old overlap (Q14), both read-only, so every p is
 * independent.  Restates windowing_short2 (block 0 against the old overlap under the previous shape's window, which also
 * clears the overlap behind it), _short3 (block 0's tail), _short4 x 7 (basic_ops.c:430-620) as what each position ends
 * up holding: positions 448 + 128 k + t hold head(k, t) + tail(k - 1, t); the last tail is left unwindowed for the next
 * frame; the first 448 positions are the old overlap at the output exponent. */
template <class Mem, class Ov>
FX_HD int32_t xu_short_sample(const Mem &x, const Ov &ov, int p, int shiftp, int shape, int shape_prev) {
  const int dd = shiftp > XU_SHIFT_OLAP ? shiftp - XU_SHIFT_OLAP : 0; /* transform side down ... */
  const int od = shiftp < XU_SHIFT_OLAP ? XU_SHIFT_OLAP - shiftp : 0; /* ... or overlap side down */
  if (p < 448) return ov[p] >> od; /* ixheaacd_scale_down(.., shift_olap, output_q), imdct.c:448 */
  if (p >= 1600) return 0;
  const int k = (p - 448) >> 7, t = (p - 448) & 127;
  const int32_t *w = xu_window(128, shape);
  int32_t head = 0, tail = 0;
  if (k < 8) {
    const int32_t *wh = k == 0 ? xu_window(128, shape_prev) : w;
    const int32_t v = t < 64 ? x[128 * k + 64 + t] : fx_neg_sat(x[128 * k + 191 - t]);
    head = xu_mul_sh1(v, wh[t]) >> dd;
  }
  if (k == 0) {
    tail = xu_mul_sh1(ov[p], xu_window(128, shape_prev)[127 - t]) >> od;
  } else {
    const int32_t v = fx_neg_sat(x[128 * (k - 1) + (t < 64 ? 63 - t : t - 64)]);
    tail = k == 8 ? (v >> dd) : (xu_mul_sh1(v, w[127 - t]) >> dd);
  }
  return k == 8 ? tail : fx_add_sat(head, tail);
}

#endif
