/*
 * usac_imdct.h -- the USAC frequency-domain IMDCT (ccfl = 1024 or 768), shared by the gfx950 kernel
 * (usac_imdct_kernel.hip) and, compiled for the host, by the checker (oracle/oracle_usac.cpp).
 *
 * Restates, as per-index functions (every butterfly of a pass and every output sample is independent, so a caller may
 * run them in any order or all at once):
 *   ixheaacd_fd_imdct_long / _short       decoder/ixheaacd_imdct.c:477 / :336
 *   ixheaacd_acelp_imdct, _fft_based_imdct, pre / post twiddle      ixheaacd_imdct.c:186 / :149 / :111 / :129
 *   ixheaacd_complex_fft_p2_dec (fft_mode = 1 branch)               ixheaacd_fft.c:1412, :1966-2484
 *   ixheaacd_complex_fft_p3 (fft_mode = 1), ixheaacd_complex_3point_fft   ixheaacd_fft.c:2531 / :2493  (ccfl 768)
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
 *   N == 512 / 128:         xu_fft_last<N>(y, b)        b = 0 .. N/2-1
 * with a barrier between passes.  Input division: xu_fft_in_shift<N>(); the exponent the reference reports:
 * xu_fft_out_shift<N>(). */
/* N = 512, 128, 64 or 16 complex points (the power-of-two transforms; 128 and 16 are the thirds of the 384- and 48-point
   ones).  fft.c:1432-1441: the input division; +1 in the reported exponent for the radix-2 pass (:2426) */
template <int N> FX_HD constexpr int xu_log2() { return N == 512 ? 9 : (N == 128 ? 7 : (N == 64 ? 6 : 4)); }
template <int N> FX_HD constexpr bool xu_not_pow4() { return (xu_log2<N>() & 1) != 0; }
template <int N> FX_HD constexpr int xu_fft_in_shift() { return xu_not_pow4<N>() ? (xu_log2<N>() + 3) / 2 : (xu_log2<N>() + 4) / 2; }
template <int N> FX_HD constexpr int xu_fft_out_shift() { return xu_fft_in_shift<N>() + (xu_not_pow4<N>() ? 1 : 0); }

template <int N, class MemIn, class Mem>
FX_HD void xu_fft_first(const MemIn &x, const Mem &y, int b) {
  constexpr int rev_shift = 15 - xu_log2<N>(); /* norm32(N) + 1 - 16 */
  unsigned h2 = xu_dig_rev((unsigned)(4 * b), rev_shift);
  if (xu_not_pow4<N>()) h2 = (h2 + 1) & ~1u;
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

/* the radix-2 pass of the 512- and 128-point transforms (fft.c:2423-2484): legs N/2 apart, twiddle step 2048 / N words */
template <int N, class Mem>
FX_HD void xu_fft_last(const Mem &y, int b) {
  const int j = b & (N / 4 - 1), p = b;
  const int32_t h = xaac_usac_fft_tw[(2048 / N) * j], l = xaac_usac_fft_tw[(2048 / N) * j + 1];
  XuCx a = {y[2 * p], y[2 * p + 1]}, c = {y[2 * (p + N / 2)], y[2 * (p + N / 2) + 1]};
  c = b < N / 4 ? xu_rot0(c, h, l) : xu_rot1(c, h, l);
  y[2 * (p + N / 2)] = fx_sub(xu_div_pow2(a.r, 1), xu_div_pow2(c.r, 1));
  y[2 * (p + N / 2) + 1] = fx_sub(xu_div_pow2(a.i, 1), xu_div_pow2(c.i, 1));
  y[2 * p] = fx_add(xu_div_pow2(a.r, 1), xu_div_pow2(c.r, 1));
  y[2 * p + 1] = fx_add(xu_div_pow2(a.i, 1), xu_div_pow2(c.i, 1));
}

/* ---- ccfl 768: the 384- and 48-point transforms = 3 x (128 | 16) points + a three-point stage (fft.c:2531) ----------
 * NT = 3 M points: sub-sequence s (s = 0, 1, 2) holds the points 3 j + s; each goes through the M-point transform above
 * (its own input division, fft.c:1443); then, per group g = 0 .. M-1 (the s-th result of each third):
 *   halve (:2575), rotate the second and third by the table (fft_mode = 1 branch, :2619-2641), three-point butterfly
 *   (:2493), and the results are points g, M + g, 2 M + g of the output (:2651-2659). */
FX_HD int32_t xu_mul32_shl(int32_t a, int32_t b) { return fx_shlw(fx_mulhi(a, b), 1); } /* ixheaac_mult32_shl */
template <int M>
FX_HD void xu_p3_group(XuCx x0, XuCx x1, XuCx x2, int g, XuCx *out /* [3]: points g, M + g, 2 M + g */) {
  const int base = g * 3 * (128 / M);
  x0.r >>= 1; x0.i >>= 1; x1.r >>= 1; x1.i >>= 1; x2.r >>= 1; x2.i >>= 1;
  {
    const int32_t wr = xaac_usac_tw3_r[base + 1], wi = xaac_usac_tw3_i[base + 1];
    const int32_t t = fx_add_sat(xu_mul_sat(x1.r, wr), xu_mul_sat(x1.i, wi));
    x1.i = fx_sub_sat(xu_mul_sat(x1.i, wr), xu_mul_sat(x1.r, wi));
    x1.r = t;
  }
  {
    const int32_t wr = xaac_usac_tw3_r[base + 2], wi = xaac_usac_tw3_i[base + 2];
    const int32_t t = fx_add_sat(xu_mul_sat(x2.r, wr), xu_mul_sat(x2.i, wi));
    x2.i = fx_sub_sat(xu_mul_sat(x2.i, wr), xu_mul_sat(x2.r, wi));
    x2.r = t;
  }
  const int32_t sinmu = -1859775393; /* sign_dir = fft_mode = 1 */
  const int32_t temp_real = fx_add_sat(x0.r, x1.r), temp_imag = fx_add_sat(x0.i, x1.i);
  const int32_t add_r = fx_add_sat(x1.r, x2.r), add_i = fx_add_sat(x1.i, x2.i);
  const int32_t sub_r = fx_sub_sat(x1.r, x2.r), sub_i = fx_sub_sat(x1.i, x2.i);
  const int32_t p1 = add_r >> 1, p4 = add_i >> 1, p2 = xu_mul32_shl(sub_i, sinmu), p3 = xu_mul32_shl(sub_r, sinmu);
  const int32_t temp = fx_sub(x0.r, p1);
  out[0].r = fx_add_sat(temp_real, x2.r);
  out[0].i = fx_add_sat(temp_imag, x2.i);
  out[1].r = fx_add_sat(temp, p2);
  out[1].i = fx_sub_sat(fx_sub_sat(x0.i, p3), p4);
  out[2].r = fx_sub_sat(temp, p2);
  out[2].i = fx_sub_sat(fx_add_sat(x0.i, p3), p4);
}
/* ixheaacd_acelp_imdct's prescale of a block whose length is not a power of two (imdct.c:197-202) */
FX_HD int32_t xu_third_twice(int32_t v) { return fx_shlw(v / 3, 1); }

/* N = complex points of a block's transform (2 N lines): 512 / 64 (ccfl 1024), 384 / 48 (ccfl 768) */
template <int N> FX_HD const int32_t *xu_pre_cos() {
  return N == 512 ? xaac_usac_pre_cos_512 : (N == 384 ? xaac_usac_pre_cos_384 : (N == 64 ? xaac_usac_pre_cos_64 : xaac_usac_pre_cos_48));
}
template <int N> FX_HD const int32_t *xu_pre_sin() {
  return N == 512 ? xaac_usac_pre_sin_512 : (N == 384 ? xaac_usac_pre_sin_384 : (N == 64 ? xaac_usac_pre_sin_64 : xaac_usac_pre_sin_48));
}
/* the power-of-two transform a block's points go through: itself, or its thirds */
template <int N> FX_HD constexpr int xu_sub_points() { return N == 384 ? 128 : (N == 48 ? 16 : N); }
/* pre twiddle of line pair i of an N-point block (imdct.c:111-127): xa = x[2i], xb = x[2N-1-2i] of the block's 2N lines
   (for ccfl 768 after the prescale); the result is already divided as the (sub-)transform's input (fft.c:1443) */
template <int N>
FX_HD XuCx xu_pre_twiddle(int32_t xa, int32_t xb, int i) {
  const int32_t c = xu_pre_cos<N>()[i], s = xu_pre_sin<N>()[i];
  XuCx y;
  y.r = fx_sub(fx_mulhi(fx_neg_sat(xa), c), fx_mulhi(xb, s));
  y.i = fx_sub(fx_mulhi(xb, c), fx_mulhi(xa, s));
  y.r = xu_div_pow2(y.r, xu_fft_in_shift<xu_sub_points<N>()>());
  y.i = xu_div_pow2(y.i, xu_fft_in_shift<xu_sub_points<N>()>());
  return y;
}
/* post twiddle (imdct.c:129-147): -> the new x[2i] (.r) and x[2N-1-2i] (.i) */
template <int N>
FX_HD XuCx xu_post_twiddle(XuCx v, int i) {
  const int32_t c = xu_pre_cos<N>()[i], s = xu_pre_sin<N>()[i];
  XuCx y;
  y.r = fx_neg(fx_sub(fx_mulhi(v.r, c), fx_mulhi(v.i, s)));
  y.i = fx_neg(fx_add(fx_mulhi(v.i, c), fx_mulhi(v.r, s)));
  return y;
}

/* what ixheaacd_acelp_imdct adds to the exponent (imdct.c:186-208): *qshift -= preshift' with
   preshift' = (reported FFT shift - preshift) + 2, preshift = log2 of the block's power-of-two part (+ 1 with the
   prescale by 2 / 3); power-of-two blocks: fft.c:2489; three-way ones: :2573 (shift by the third's size, + 1) */
template <int N> FX_HD constexpr int xu_imdct_q_gain() {
  return N == 512 ? 10 - 7 - 2 : (N == 64 ? 7 - 5 - 2 : (N == 384 ? 9 - (6 + 1) - 2 : /* 48 */ 6 - (4 + 1) - 2));
}

/* ixheaacd_normalize with the count the second renormalisation can reach (imdct.c:94-100, :517): max_shift - 1 is -1
   when the transform's peak already fills the word -- a negative shift count, undefined in C; here: count & 31 */
FX_HD int32_t xu_normalize(int32_t v, int shift) { return fx_shlw(v, shift & 31); }

FX_HD const int32_t *xu_window(int len, int shape) { /* ixheaacd_calc_window (ixheaacd_Windowing.c:29) */
  switch (len) {
    case 1024: return shape ? xaac_usac_kbd_win_1024 : xaac_usac_sine_win_1024;
    case 768: return shape ? xaac_usac_kbd_win_768 : xaac_usac_sine_win_768;
    case 128: return shape ? xaac_usac_kbd_win_128 : xaac_usac_sine_win_128;
    case 256: return xaac_usac_sine_win_256; /* no KBD window of this length: xu_lpd_window_missing() */
    case 192: return shape ? xaac_usac_kbd_win_192 : xaac_usac_sine_win_192;
    default: return shape ? xaac_usac_kbd_win_96 : xaac_usac_sine_win_96;
  }
}

/* ---- the frame behind an LPD frame (td_frame_prev) and forward-aliasing cancellation (fac_data_present) ---------------
 * ixheaacd_fd_frm_dec (imdct.c:618-633): behind a time-domain frame the left slope is 2 lfac samples long with
 * lfac = ccfl / 16 (EIGHT_SHORT) or ccfl / 8 (other sequences); otherwise it is the short window's.  FAC data, when
 * present, is the 2 lfac-sample signal ixheaacd_cal_fac_data leaves in fac_idata (with its exponent fac_q): computed by
 * the LPD decoder's side (LPC synthesis of the transmitted FAC spectrum + the ACELP zero-input response) and handed in. */
struct XuLpd {
  int td_prev, fac;     /* usac_data->td_frame_prev, ->fac_data_present */
  int fac_q;            /* exponent of the FAC signal */
};
template <int L> FX_HD constexpr int xu_lfac(bool td_prev, bool eight_short) { return td_prev ? (eight_short ? L / 16 : L / 8) : 128; /* FAC_LENGTH */ }
/* ixheaacd_calc_window fails for a KBD window of 256 taps (ixheaacd_Windowing.c:63-104): ccfl 1024, LONG_STOP / STOP_START
   behind an LPD frame with window_shape_prev = 1 */
template <int L> FX_HD bool xu_lpd_window_missing(bool td_prev, int seq, int shape_prev) {
  return L == 1024 && td_prev && (seq == 3 || seq == 4) && shape_prev == 1;
}

/* fac_q comes from the caller (ixheaacd_cal_fac_data's *q_fac = qshift1 - preshift, imdct.c:325): every shift count the FAC
   windowing forms from it and the frame's transform exponent must lie in 0..31 -- outside, the reference's own C shifts are
   undefined, and a CPU and a GPU would silently disagree.  Such a frame is refused (XAAC_FATAL_BAD_ARG, nothing written).
   x_zero: the transform's output is all zeros (a silent frame: shiftp sits at its cap of 31 + XU_SHIFT_OLAP and the
   transform side's count may pass 31 -- shifting zeros, which every machine answers with zero; the reference-made chains
   hold such frames). */
FX_HD bool xu_fac_q_ok(int shiftp, bool x_zero, bool eight_short, int fac_q) {
  if (eight_short) { /* ixheaacd_combine_fac (basic_ops.c:56) at the short path's output exponent */
    const int d = fac_q - (shiftp > XU_SHIFT_OLAP ? XU_SHIFT_OLAP : shiftp);
    return d >= -31 && d <= 31;
  }
  int q = shiftp < XU_SHIFT_OLAP ? shiftp : XU_SHIFT_OLAP; /* windowing_long2 (basic_ops.c:121) + ixheaacd_scale_down_adj */
  q = fac_q < q ? fac_q : q;
  return (x_zero || shiftp - q <= 31) && XU_SHIFT_OLAP - q <= 31 && fac_q - q <= 31 && 15 - q <= 31;
}

/* ---- long blocks: output sample i (0 .. L-1) of the frame, before the final rescale; L = ccfl -----------------------
 * x: the L transform outputs after the second renormalisation; ov: the overlap (Q14); shiftp: their exponent.
 * ONLY_LONG / LONG_START: windowing_long1 (basic_ops.c:77); LONG_STOP / STOP_START: windowing_long3 (:298), no FAC:
 * flat part of (L - L/8) / 2 samples, then the L/8-sample slope of the previous shape's short window. */
template <int L, class Mem, class Ov, class Fac>
FX_HD int32_t xu_long_sample_lpd(const Mem &x, const Ov &ov, const Fac &fac, int i, int shiftp, bool stop_like, int shape_prev,
                                 const XuLpd &lp) {
  constexpr int H = L / 2;
  if (!stop_like) { /* windowing_long1: the whole frame under the previous shape's long window */
    const int d = shiftp - XU_SHIFT_OLAP; /* > 0: the transform side is shifted down; <= 0: the overlap side */
    const int32_t *w = xu_window(L, shape_prev);
    const int m = i < H ? i : L - 1 - i;              /* the loop index of basic_ops.c:85 */
    const int32_t src1 = x[H + m];
    const int32_t t = i < H ? xu_mul_sh1(src1, w[m]) : xu_mul_sh1(fx_neg_sat(src1), w[L - 1 - m]);
    const int32_t o = i < H ? xu_mul_sh1(ov[m], w[L - 1 - m]) : xu_mul_sh1(ov[L - 1 - m], w[m]);
    return d > 0 ? fx_add_sat(t >> d, o) : fx_add_sat(t, o >> -d);
  }
  /* LONG_STOP / STOP_START: flat part F, slope of W samples under the previous shape's window of that length */
  const int lfac = xu_lfac<L>(lp.td_prev != 0, false);
  const int W = lp.td_prev ? 2 * lfac : L / 8, F = (L - W) / 2;
  const int32_t *w = xu_window(W, shape_prev);
  if (!lp.fac) { /* windowing_long3 (basic_ops.c:298) */
    const int d = shiftp - XU_SHIFT_OLAP;
    if (i < F) return d > 0 ? ov[i] : (ov[i] >> -d);
    if (i < F + W) {
      const int32_t src = i < H ? x[H + i] : fx_neg_sat(x[H + L - 1 - i]);
      const int32_t t = xu_mul_sh1(src, w[i - F]), o = xu_mul_sh1(ov[i], w[W - 1 - (i - F)]);
      return d > 0 ? fx_add_sat(t >> d, o) : fx_add_sat(t, o >> -d);
    }
    const int32_t v = fx_neg_sat(x[H + L - 1 - i]);
    return d > 0 ? (v >> d) : v;
  }
  /* windowing_long2 (basic_ops.c:121): the old overlap up to F + lfac, the second half of the slope, the FAC signal over
     [F + lfac, F + 3 lfac); every term comes down to q = the smallest of the three exponents (the four branches of the
     reference are this rule; a shift by 0 is the identity) */
  int q = shiftp < XU_SHIFT_OLAP ? shiftp : XU_SHIFT_OLAP;
  q = lp.fac_q < q ? lp.fac_q : q;
  const int dx = shiftp - q, dov = XU_SHIFT_OLAP - q, df = lp.fac_q - q;
  if (i < F + lfac) return ov[i] >> dov;
  const int32_t v = fx_neg_sat(x[H + L - 1 - i]);     /* i >= F + lfac = L / 2 */
  if (i < F + W) return fx_add_sat(xu_mul_sh1(v, w[i - F]) >> dx, fac[i - F - lfac] >> df);
  if (i < F + 3 * lfac) return fx_add_sat(v >> dx, fac[i - F - lfac] >> df);
  return v >> dx;
}
/* the exponent xu_long_sample_lpd's value has */
FX_HD int xu_long_output_q_lpd(int shiftp, bool stop_like, const XuLpd &lp) {
  int q = shiftp > XU_SHIFT_OLAP ? XU_SHIFT_OLAP : shiftp;
  if (stop_like && lp.fac && lp.fac_q < q) q = lp.fac_q;
  return q;
}
/* imdct.c:459-470: p_out_buffer = (FLOAT32)out * 2^-15; [bass post filter]; out = (WORD32)(p_out_buffer * 32768) */
FX_HD int32_t xu_float_round_trip(int32_t v) { return fx_f2i_trunc(((float)v * 0.000030517578125f) * 32768.0f); }
struct XuNoFac {
  FX_MEMBER int32_t operator[](int) const { return 0; }
};
template <int L, class Mem, class Ov>
FX_HD int32_t xu_long_sample(const Mem &x, const Ov &ov, int i, int shiftp, bool stop_like, int shape_prev) {
  const XuLpd none = {0, 0, 0};
  return xu_long_sample_lpd<L>(x, ov, XuNoFac(), i, shiftp, stop_like, shape_prev, none);
}
FX_HD int xu_long_output_q(int shiftp) { return shiftp > XU_SHIFT_OLAP ? XU_SHIFT_OLAP : shiftp; }
/* the new overlap, sample i (imdct.c:563-576: both branches shift right) */
template <int L, class Mem>
FX_HD int32_t xu_long_overlap(const Mem &x, int i, int shiftp) {
  const int d = shiftp > XU_SHIFT_OLAP ? shiftp - XU_SHIFT_OLAP : XU_SHIFT_OLAP - shiftp;
  const int m = i >= L / 2 ? i - L / 2 : L / 2 - 1 - i;
  return fx_neg_sat(x[m]) >> d;
}
/* ixheaacd_scale_down_adj(.., output_q, 15) (basic_ops.c:640): the frame's Q15 output */
FX_HD int32_t xu_scale_adj(int32_t v, int output_q) {
  return fx_add_sat(output_q > 15 ? (v >> (output_q - 15)) : fx_shl_sat(v, 15 - output_q), 11);
}
/* ixheaacd_scale_down (basic_ops.c:622) */
FX_HD int32_t xu_scale(int32_t v, int from_q, int to_q) { return from_q > to_q ? (v >> (from_q - to_q)) : fx_shl_sat(v, to_q - from_q); }

/* ---- EIGHT_SHORT frames: sample p (0 .. 2L-1) of the reference's overlap_data_buf after the eight windowed blocks ------
 * x: the 8 x L/8 transform outputs after the renormalisation, ov: the old overlap (Q14), both read-only, so every p is
 * independent.  Restates windowing_short2 (block 0 against the old overlap under the previous shape's window, which also
 * clears the overlap behind it), _short3 (block 0's tail), _short4 x 7 (basic_ops.c:430-620) as what each position ends
 * up holding: with S = L/8, F = (L - S) / 2 positions F + S k + t hold head(k, t) + tail(k - 1, t); the last tail is left
 * unwindowed for the next frame; the first F positions are the old overlap at the output exponent. */
template <int L, class Mem, class Ov, class Fac>
FX_HD int32_t xu_short_sample_lpd(const Mem &x, const Ov &ov, const Fac &fac, int p, int shiftp, int shape, int shape_prev,
                                  const XuLpd &lp) {
  constexpr int S = L / 8, F = (L - S) / 2; /* behind an LPD frame lfac = L / 16: the same flat part and slope length */
  const int dd = shiftp > XU_SHIFT_OLAP ? shiftp - XU_SHIFT_OLAP : 0; /* transform side down ... */
  const int od = shiftp < XU_SHIFT_OLAP ? XU_SHIFT_OLAP - shiftp : 0; /* ... or overlap side down */
  const int lfac = xu_lfac<L>(lp.td_prev != 0, true);
  if (p < F) return ov[p] >> od; /* ixheaacd_scale_down(.., shift_olap, output_q), imdct.c:448 */
  if (p >= F + 9 * S) return 0;
  const int k = (p - F) / S, t = (p - F) % S;
  const int32_t *w = xu_window(S, shape);
  int32_t v;
  if (lp.fac && k == 0) {
    /* windowing_short1 (basic_ops.c:373): up to lfac the old overlap alone, then block 0's falling half under the
       previous window WITHOUT the old overlap (and nothing where lfac reaches past the block) */
    if (t < lfac) v = ov[p] >> od;
    else v = xu_mul_sh1(fx_neg_sat(x[S + S / 2 - 1 - t]), xu_window(S, shape_prev)[t]) >> dd;
  } else {
    int32_t head = 0, tail = 0;
    if (k < 8) {
      const int32_t *wh = k == 0 ? xu_window(S, shape_prev) : w;
      const int32_t hv = t < S / 2 ? x[S * k + S / 2 + t] : fx_neg_sat(x[S * k + S + S / 2 - 1 - t]);
      head = xu_mul_sh1(hv, wh[t]) >> dd;
    }
    if (k == 0) {
      tail = xu_mul_sh1(ov[p], xu_window(S, shape_prev)[S - 1 - t]) >> od;
    } else {
      const int32_t tv = fx_neg_sat(x[S * (k - 1) + (t < S / 2 ? S / 2 - 1 - t : t - S / 2)]);
      tail = k == 8 ? (tv >> dd) : (xu_mul_sh1(tv, w[S - 1 - t]) >> dd);
      /* with FAC windowing_short1 clears the old overlap only up to 2 F + lfac (it clears n_flat_ls + lfac words from the
         block on): what lies between there and 2 F + S is still in place when a later block's tail is added onto it */
      if (lp.fac && p >= 2 * F + lfac && p < 2 * F + S) tail = fx_add_sat(tail, ov[p] >> od);
    }
    v = k == 8 ? tail : fx_add_sat(head, tail);
  }
  if (lp.fac && p >= F + lfac && p < F + 3 * lfac) { /* ixheaacd_combine_fac (basic_ops.c:56) */
    const int oq = shiftp > XU_SHIFT_OLAP ? XU_SHIFT_OLAP : shiftp;
    const int32_t f = fac[p - F - lfac];
    v = fx_add_sat(v, lp.fac_q > oq ? (f >> (lp.fac_q - oq)) : fx_shl_sat(f, oq - lp.fac_q));
  }
  return v;
}
template <int L, class Mem, class Ov>
FX_HD int32_t xu_short_sample(const Mem &x, const Ov &ov, int p, int shiftp, int shape, int shape_prev) {
  const XuLpd none = {0, 0, 0};
  return xu_short_sample_lpd<L>(x, ov, XuNoFac(), p, shiftp, shape, shape_prev, none);
}

#endif
