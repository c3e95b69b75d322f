/*
 * imdct960.h -- the 960-line AAC inverse transform (frame_length 960: one 960-line block or eight 120-line blocks) with
 * its windowing / overlap-add, shared by the gfx950 kernel (imdct960_kernel.hip) and, compiled for the host, by the
 * checker (oracle/oracle_imdct960.cpp).
 *
 * Restates the frame_length == 960 branches of ixheaacd_imdct_process (decoder/ixheaacd_lpfuncs.c:347-802):
 *   ixheaacd_mdct_960 / ixheaacd_inverse_transform_960          decoder/ixheaacd_aac_imdct.c:1672 / :1624
 *   ixheaacd_pre_twiddle_960 / _120, ixheaacd_post_twiddle_960 / _120   aac_imdct.c:2489 / :2533 / :2704 / :2664
 *   ixheaacd_fft_960 = 15 x ixheaacd_fft_32_points + 32 x ixheaacd_ld_dec_fft_15_opt   aac_imdct.c:1792 / :1823 / :1975
 *   ixheaacd_fft_120 = 15 x 4-point + 4 x ixheaacd_fft_960_15 (ixheaacd_fft_5, _fft_3)  aac_imdct.c:2253 / :2334 / :2398 / :2366
 *   ixheaacd_over_lap_add1_dec / _add2_dec (block.c:1193 / :1220), ixheaacd_process_win_seq, ixheaacd_long_short_win_seq,
 *   ixheaacd_nolap1_32, ixheaacd_Nolap_dec, ixheaacd_spec_to_overlapbuf_dec, ixheaacd_overlap_buf_out_dec,
 *   ixheaacd_overlap_out_copy_dec (lpfuncs.c:94-346), ixheaacd_dec_copy_outsample (block.c:1131)
 *
 * Form: every stage is a loop over independent work items written once with X9_FOR (item = lane, lane + nl, ...):
 * the host runs it with (lane, nl) = (0, 1), a wave with (lane, 64) and x9_sync() between stages.  The reference's index
 * tables (prime-factor input / output maps) are read where the data is, so no stage only moves data.  The arithmetic is
 * the reference's operation for operation: saturating adds in the FFTs, wrapping sums and negations in the pre twiddle,
 * (a * b) >> 16 products with the twiddles (32-bit words holding Q15 values for 960, 16-bit ones for 120).
 */
#ifndef XAAC_IMDCT960_H
#define XAAC_IMDCT960_H

#include "fx.h"

#ifndef XAAC_I960_TABLES_INCLUDED
#define XAAC_I960_TABLES_INCLUDED
#if defined(__HIPCC__)
#define XAAC_TAB_QUAL static __device__ const
#include "tables_imdct960.inc"
#undef XAAC_TAB_QUAL
#else
#include "tables_imdct960.inc"
#endif
#endif

#define X9_FOR(i, n) for (int i = lane; i < (n); i += nl)

enum { X9_ONLY_LONG = 0, X9_LONG_START = 1, X9_EIGHT_SHORT = 2, X9_LONG_STOP = 3 };

/* between stages: a wave-level barrier on the device, nothing on the host */
FX_HD void x9_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

struct X9Cx {
  int32_t r, i;
};

FX_HD X9Cx x9_ld(const int32_t *p, int c) {
  X9Cx v = {p[2 * c], p[2 * c + 1]};
  return v;
}
FX_HD void x9_st(int32_t *p, int c, X9Cx v) {
  p[2 * c] = v.r;
  p[2 * c + 1] = v.i;
}
FX_HD X9Cx x9_add(X9Cx a, X9Cx b) {
  X9Cx v = {fx_add_sat(a.r, b.r), fx_add_sat(a.i, b.i)};
  return v;
}
FX_HD X9Cx x9_sub(X9Cx a, X9Cx b) {
  X9Cx v = {fx_sub_sat(a.r, b.r), fx_sub_sat(a.i, b.i)};
  return v;
}

/* ixheaac_shr32_dir_sat (basic_ops32.h:114): b < 0 -> saturating left shift */
FX_HD int32_t x9_shr_dir_sat(int32_t a, int b) { return b < 0 ? fx_shl_sat(a, -b) : fx_shr(a, b); }

/* MPYLIRC (aac_imdct.c:61): 16-bit coefficient times 32-bit sample, low half rounded */
FX_HD int32_t x9_mpylirc(int16_t x, int32_t y) {
  const int32_t lo = ((int32_t)x * (int32_t)(uint16_t)(y & 0xffff) + 0x4000) >> 15;
  const int32_t hi = fx_shlw((int32_t)x * (int32_t)(int16_t)(y >> 16), 1);
  return fx_add(lo, hi);
}

/* ---- pre twiddle: complex element c of the n / 2 the FFT takes (aac_imdct.c:2489 / :2533) ---------------- */
template <bool LONG>
FX_HD int32_t x9_tw_mul(int32_t a, int k) {
  /* ixheaac_mult32x32in32 (basic_ops40.h:45) is (a * b) >> 16 cut to 32 bits, and cosine_array_1920's 32-bit words hold Q15
     values (the table generator asserts it): the same product as with the 16-bit table of the short transform */
  return fx_mul32x16(a, LONG ? xaac_i960_cos_1920[k] : xaac_i960_cos_240[k]);
}

template <bool LONG>
FX_HD X9Cx x9_pre_twiddle(const int32_t *data, int c, int sh) {
  constexpr int n = LONG ? 960 : 120;
  X9Cx v;
  if (c < n / 4) {
    const int i = c;
    const int32_t tr = data[2 * i], ti = data[n - 1 - 2 * i];
    v.r = x9_shr_dir_sat(fx_neg(fx_add(x9_tw_mul<LONG>(tr, 4 * i), x9_tw_mul<LONG>(ti, 4 * i + 1))), sh);
    v.i = x9_shr_dir_sat(fx_neg(fx_sub(x9_tw_mul<LONG>(ti, 4 * i), x9_tw_mul<LONG>(tr, 4 * i + 1))), sh);
  } else {
    const int i = n / 2 - 1 - c;
    const int32_t ti = data[2 * i + 1], tr = data[n - 2 - 2 * i];
    v.i = x9_shr_dir_sat(fx_neg(fx_sub(x9_tw_mul<LONG>(ti, 4 * i + 2), x9_tw_mul<LONG>(tr, 4 * i + 3))), sh);
    v.r = x9_shr_dir_sat(fx_neg(fx_add(x9_tw_mul<LONG>(tr, 4 * i + 2), x9_tw_mul<LONG>(ti, 4 * i + 3))), sh);
  }
  return v;
}

/* ---- post twiddle + the 17476 scale: pair k -> out[2k], out[n-1-2k], out[2k+1], out[n-2-2k]  (:2704 / :2664) -- */
FX_HD int32_t x9_scale(int32_t a) { return fx_mul32x16_shl(a, 17476); }

template <bool LONG>
FX_HD void x9_post_twiddle(X9Cx lo, X9Cx hi, int k, int32_t *out) {
  constexpr int n = LONG ? 960 : 120;
  out[n - 1 - 2 * k] = x9_scale(fx_neg(fx_sub_sat(x9_tw_mul<LONG>(lo.r, 4 * k + 1), x9_tw_mul<LONG>(lo.i, 4 * k))));
  out[2 * k] = x9_scale(fx_neg(fx_add_sat(x9_tw_mul<LONG>(lo.r, 4 * k), x9_tw_mul<LONG>(lo.i, 4 * k + 1))));
  out[2 * k + 1] = x9_scale(fx_neg(fx_sub_sat(x9_tw_mul<LONG>(hi.r, 4 * k + 3), x9_tw_mul<LONG>(hi.i, 4 * k + 2))));
  out[n - 2 - 2 * k] = x9_scale(fx_neg(fx_add_sat(x9_tw_mul<LONG>(hi.r, 4 * k + 2), x9_tw_mul<LONG>(hi.i, 4 * k + 3))));
}

/* ---- 5- and 3-point transforms (aac_imdct.c:2398 / :2366; inside ixheaacd_ld_dec_fft_15_opt :1975 the same two, but its
   5-point stage doubles with a wrapping << 1 where ixheaacd_fft_5 saturates) ---- */
template <bool SAT>
FX_HD int32_t x9_dbl(int32_t a) {
  return SAT ? fx_shl_sat(a, 1) : fx_shlw(a, 1);
}

template <bool SAT>
FX_HD void x9_fft5(const X9Cx x[5], X9Cx y[5]) {
  const int32_t c_51 = 2042378317, c_52 = -1652318768, c_53 = -780119100, c_54 = 1200479854, c_55 = -1342177280;
  int32_t r1 = fx_add_sat(x[1].r, x[4].r), r4 = fx_sub_sat(x[1].r, x[4].r);
  int32_t r3 = fx_add_sat(x[2].r, x[3].r), r2 = fx_sub_sat(x[2].r, x[3].r);
  int32_t t = fx_mul32_shl(fx_sub_sat(r1, r3), c_54);
  r1 = fx_add_sat(r1, r3);
  const int32_t temp1 = fx_add_sat(x[0].r, r1);
  r1 = fx_add_sat(temp1, x9_dbl<SAT>(fx_mul32_shl(r1, c_55)));
  r3 = fx_sub_sat(r1, t);
  r1 = fx_add_sat(r1, t);
  t = fx_mul32_shl(fx_add_sat(r4, r2), c_51);
  r4 = fx_add_sat(t, x9_dbl<SAT>(fx_mul32_shl(r4, c_52)));
  r2 = fx_add_sat(t, fx_mul32_shl(r2, c_53));
  int32_t s1 = fx_add_sat(x[1].i, x[4].i), s4 = fx_sub_sat(x[1].i, x[4].i);
  int32_t s3 = fx_add_sat(x[2].i, x[3].i), s2 = fx_sub_sat(x[2].i, x[3].i);
  t = fx_mul32_shl(fx_sub_sat(s1, s3), c_54);
  s1 = fx_add_sat(s1, s3);
  const int32_t temp2 = fx_add_sat(x[0].i, s1);
  s1 = fx_add_sat(temp2, x9_dbl<SAT>(fx_mul32_shl(s1, c_55)));
  s3 = fx_sub_sat(s1, t);
  s1 = fx_add_sat(s1, t);
  t = fx_mul32_shl(fx_add_sat(s4, s2), c_51);
  s4 = fx_add_sat(t, x9_dbl<SAT>(fx_mul32_shl(s4, c_52)));
  s2 = fx_add_sat(t, fx_mul32_shl(s2, c_53));
  y[0].r = temp1;
  y[0].i = temp2;
  y[1].r = fx_add_sat(r1, s2);
  y[1].i = fx_sub_sat(s1, r2);
  y[2].r = fx_sub_sat(r3, s4);
  y[2].i = fx_add_sat(s3, r4);
  y[3].r = fx_add_sat(r3, s4);
  y[3].i = fx_sub_sat(s3, r4);
  y[4].r = fx_sub_sat(r1, s2);
  y[4].i = fx_add_sat(s1, r2);
}

FX_HD void x9_fft3(X9Cx x0, X9Cx x1, X9Cx x2, X9Cx y[3]) {
  const int32_t sinmu = 1859775393;
  const int32_t x01r = fx_add_sat(x0.r, x1.r), x01i = fx_add_sat(x0.i, x1.i);
  const int32_t add_r = fx_add_sat(x1.r, x2.r), add_i = fx_add_sat(x1.i, x2.i);
  const int32_t sub_r = fx_sub_sat(x1.r, x2.r), sub_i = fx_sub_sat(x1.i, x2.i);
  const int32_t p1 = add_r >> 1, p2 = fx_mul32_shl(sub_i, sinmu), p3 = fx_mul32_shl(sub_r, sinmu), p4 = add_i >> 1;
  const int32_t temp = fx_sub_sat(x0.r, p1);
  y[0].r = fx_add_sat(x01r, x2.r);
  y[0].i = fx_add_sat(x01i, x2.i);
  y[1].r = fx_add_sat(temp, p2);
  y[2].r = fx_sub_sat(temp, p2);
  y[1].i = fx_sub_sat(fx_sub_sat(x0.i, p3), p4);
  y[2].i = fx_sub_sat(fx_add_sat(x0.i, p3), p4);
}

/* ---- the 32-point transform's radix-4 butterfly (aac_imdct.c:1862-1912): positions p, p+h, p+2h, p+3h ------- */
FX_HD void x9_fft32_bfly(int32_t *x, int p, int h, const int16_t *w) {
  const X9Cx a = x9_ld(x, p), b = x9_ld(x, p + h), c = x9_ld(x, p + 2 * h), d = x9_ld(x, p + 3 * h);
  const int16_t si10 = w[0], co10 = w[1], si20 = w[2], co20 = w[3], si30 = w[4], co30 = w[5];
  const X9Cx xh = x9_add(a, c), xl = x9_sub(a, c), xh2 = x9_add(b, d), xl2 = x9_sub(b, d);
  x9_st(x, p, x9_add(xh, xh2));
  const int32_t xt0 = fx_sub_sat(xh.r, xh2.r), yt0 = fx_sub_sat(xh.i, xh2.i);
  const int32_t xt1 = fx_add_sat(xl.r, xl2.i), yt2 = fx_add_sat(xl.i, xl2.r);
  const int32_t xt2 = fx_sub_sat(xl.r, xl2.i), yt1 = fx_sub_sat(xl.i, xl2.r);
  X9Cx v;
  v.r = fx_add_sat(x9_mpylirc(si10, yt1), x9_mpylirc(co10, xt1));
  v.i = fx_sub_sat(x9_mpylirc(co10, yt1), x9_mpylirc(si10, xt1));
  x9_st(x, p + h, v);
  v.r = fx_add_sat(x9_mpylirc(si20, yt0), x9_mpylirc(co20, xt0));
  v.i = fx_sub_sat(x9_mpylirc(co20, yt0), x9_mpylirc(si20, xt0));
  x9_st(x, p + 2 * h, v);
  v.r = fx_add_sat(x9_mpylirc(si30, yt2), x9_mpylirc(co30, xt2));
  v.i = fx_sub_sat(x9_mpylirc(co30, yt2), x9_mpylirc(si30, xt2));
  x9_st(x, p + 3 * h, v);
}

/* ---- one 960-line block: spec -> y (both 960 words), a = 960 words of work space; returns q_shift ----------- */
/* buffers: y and a may be LDS; spec is only read.  e = headroom(spec) - 1 (aac_imdct.c:1679). */
FX_HD int x9_long_transform(const int32_t *spec, int32_t *y, int32_t *a, int e, int lane, int nl) {
  const int sh = 7 - e;
  /* pre twiddle through the first index table: y[n] = z[re_arr_tab_32[n]]  (:1798) */
  X9_FOR(n, 480) x9_st(y, n, x9_pre_twiddle<true>(spec, xaac_i960_arr_32[n], sh));
  x9_sync();
  /* 15 x 32 points, in place: 8 butterflies of span 8, then 4 x 2 of span 2 (:1846-1912) */
  X9_FOR(t, 120) x9_fft32_bfly(y, 32 * (t >> 3) + (t & 7), 8, xaac_i960_w_32 + 6 * (t & 7));
  x9_sync();
  X9_FOR(t, 120) x9_fft32_bfly(y, 32 * (t >> 3) + 8 * ((t >> 1) & 3) + (t & 1), 2, xaac_i960_w_32 + 48 + 6 * (t & 1));
  x9_sync();
  /* last radix-2 pass with its digit reversal (:1914-1961): item = (block, i, half) -> four outputs */
  X9_FOR(t, 120) {
    const int b = t >> 3, i = (t >> 1) & 3, half = t & 1;
    const int rev = (i == 0 ? 0 : i == 1 ? 8 : i == 2 ? 2 : 10) + half;
    const int src = 32 * b + (i < 2 ? 4 * i : 16 + 4 * (i - 2)) + 8 * half;
    const X9Cx x0 = x9_ld(y, src), x1 = x9_ld(y, src + 1), x2 = x9_ld(y, src + 2), x3 = x9_ld(y, src + 3);
    x9_st(a, 32 * b + rev, x9_add(x0, x1));
    x9_st(a, 32 * b + 16 + rev, x9_sub(x0, x1));
    x9_st(a, 32 * b + 4 + rev, x9_add(x2, x3));
    x9_st(a, 32 * b + 20 + rev, x9_sub(x2, x3));
  }
  x9_sync();
  /* 32 x 15 points (:1975): three 5-point transforms on inputs 64 words apart in the reference's order ... */
  X9_FOR(t, 96) {
    const int j = t & 31, g = t >> 5;
    X9Cx in[5], out[5];
#pragma unroll
    for (int m = 0; m < 5; m++) {
      const int k = 5 * g + 3 * m; /* (5 g + 3 m) mod 15 */
      in[m] = x9_ld(a, j + 32 * (k >= 15 ? k - 15 : k));
    }
    x9_fft5<false>(in, out);
#pragma unroll
    for (int m = 0; m < 5; m++) x9_st(y, 15 * j + 5 * g + m, out[m]);
  }
  x9_sync();
  /* ... then five 3-point ones, results through re_arr_tab_sml_480 */
  X9_FOR(t, 160) {
    const int j = t & 31, i = t >> 5;
    X9Cx out[3];
    x9_fft3(x9_ld(y, 15 * j + i), x9_ld(y, 15 * j + 5 + i), x9_ld(y, 15 * j + 10 + i), out);
#pragma unroll
    for (int m = 0; m < 3; m++) x9_st(a, xaac_i960_arr_sml_480[15 * j + 3 * i + m], out[m]);
  }
  x9_sync();
  X9_FOR(k, 240) x9_post_twiddle<true>(x9_ld(a, k), x9_ld(a, 479 - k), k, y);
  x9_sync();
  return 15 - e;
}

/* ---- eight 120-line blocks: spec -> y, a = 960 words of work space; returns q_shift (:1624, lpfuncs.c:684-698) -- */
FX_HD int x9_short_transform(const int32_t *spec, int32_t *y, int32_t *a, int e, int lane, int nl) {
  const int sh = 4 - e;
  /* pre twiddle through re_arr_tab_4 (:2262) */
  X9_FOR(t, 480) {
    const int w = t & 7, n = t >> 3;
    x9_st(y, 60 * w + n, x9_pre_twiddle<false>(spec + 120 * w, xaac_i960_arr_4[n], sh));
  }
  x9_sync();
  /* fifteen 4-point transforms per block (:2266-2316) */
  X9_FOR(t, 120) {
    const X9Cx x0 = x9_ld(y, 4 * t), x1 = x9_ld(y, 4 * t + 1), x2 = x9_ld(y, 4 * t + 2), x3 = x9_ld(y, 4 * t + 3);
    const X9Cx xh0 = x9_add(x0, x2), xl0 = x9_sub(x0, x2), xh1 = x9_add(x1, x3), xl1 = x9_sub(x1, x3);
    X9Cx n1, n3;
    n1.r = fx_add_sat(xl0.r, xl1.i);
    n1.i = fx_sub_sat(xl0.i, xl1.r);
    n3.r = fx_sub_sat(xl0.r, xl1.i);
    n3.i = fx_add_sat(xl0.i, xl1.r);
    x9_st(a, 4 * t, x9_add(xh0, xh1));
    x9_st(a, 4 * t + 1, n1);
    x9_st(a, 4 * t + 2, x9_sub(xh0, xh1));
    x9_st(a, 4 * t + 3, n3);
  }
  x9_sync();
  /* four 15-point transforms per block (:2334): inputs through re_arr_tab_15_4 and re_arr_tab_5 (composed by the table
     generator); 5-point stage */
  X9_FOR(t, 96) {
    const int wb = t & 31, g = t >> 5; /* wb = 4 w + b: 15-point transform wb starts at complex 15 wb */
    const int32_t *src = a + 2 * 60 * (wb >> 2);
    X9Cx in[5], out[5];
#pragma unroll
    for (int m = 0; m < 5; m++) in[m] = x9_ld(src, xaac_i960_arr_15_4_5[15 * (wb & 3) + 5 * g + m]);
    x9_fft5<true>(in, out);
#pragma unroll
    for (int m = 0; m < 5; m++) x9_st(y, 15 * wb + 5 * g + m, out[m]);
  }
  x9_sync();
  /* 3-point stage through re_arr_tab_3 */
  X9_FOR(t, 160) {
    const int wb = t & 31, g = t >> 5;
    const int32_t *src = y + 2 * 15 * wb;
    X9Cx out[3];
    x9_fft3(x9_ld(src, xaac_i960_arr_3[3 * g]), x9_ld(src, xaac_i960_arr_3[3 * g + 1]), x9_ld(src, xaac_i960_arr_3[3 * g + 2]), out);
#pragma unroll
    for (int m = 0; m < 3; m++) x9_st(a, 15 * wb + 3 * g + m, out[m]);
  }
  x9_sync();
  /* post twiddle reading through re_arr_tab_sml and re_arr_tab_120 (:2362, :2331; the two composed by the table generator) */
  X9_FOR(t, 240) {
    const int w = t & 7, k = t >> 3;
    const X9Cx lo = x9_ld(a, 60 * w + xaac_i960_arr_120_sml[k]), hi = x9_ld(a, 60 * w + xaac_i960_arr_120_sml[59 - k]);
    x9_post_twiddle<false>(lo, hi, k, y + 120 * w);
  }
  x9_sync();
  return 15 - e;
}

/* ---- windowing / overlap-add, u = 60: the reference's helpers with its size_01 argument ------------------------ */
#define X9_U 60

/* where a finished time sample goes */
struct X9Sink {
  int32_t *o32; /* or null */
  int16_t *p16; /* or null */
  int stride;
  int qadj;
  int mode; /* 0: x * 2^qadj wrapping, round16 (LC hand-off); 1: round16(shl32_sat(x, qadj)) (SBR hand-off) */
  FX_MEMBER void put(int n, int32_t v) const {
    if (o32) o32[n * stride] = v;
    if (p16) p16[n * stride] = fx_round16(mode ? fx_shl_sat(v, qadj) : fx_shlw(v, qadj));
  }
};

FX_HD const int16_t *x9_long_win(int shape) { return shape ? xaac_i960_win_long_kbd : xaac_i960_win_long_sine; }
FX_HD const int16_t *x9_short_win(int shape) { return shape ? xaac_i960_win_short_kbd : xaac_i960_win_short_sine; }

/* lpfuncs.c:316 */
FX_HD int32_t x9_to_ovl(int32_t v, int q) { return fx_shr_rnd(v, 16 - q); }

/* block.c:1193: coef -> 2n block (upper half read), prev -> n old-overlap words */
FX_HD void x9_ola1(const int32_t *coef, const int32_t *prev, const X9Sink &sk, int obase, const int16_t *win, int q, int n,
                   int lane, int nl) {
  X9_FOR(i, n) {
    const int16_t w1 = win[2 * n - 2 * i - 1], w2 = win[2 * n - 2 * i - 2];
    const int32_t c = coef[2 * n - 1 - i], p = prev[i];
    sk.put(obase + n - 1 - i, fx_sub_sat(fx_shl_dir_sat_limit(fx_mul32x16(c, w2), q), fx_mul32x16_nosh_sat(p, w1)));
    sk.put(obase + n + i, fx_sub_sat(fx_shl_dir_sat_limit(fx_mul32x16(fx_neg_sat(c), w1), q), fx_mul32x16_nosh_sat(p, w2)));
  }
}

/* block.c:1220: value i (0 .. 2n-1) of a short / short overlap */
FX_HD int32_t x9_ola2_value(const int32_t *coef, const int32_t *prev, const int16_t *win, int q, int n, int i) {
  int32_t a;
  if (i < n) {
    a = fx_sub_sat(fx_mul32x16(coef[n + i], win[2 * i]), fx_mul32x16(prev[n - 1 - i], win[2 * i + 1]));
  } else {
    const int j = i - n;
    a = fx_sub_sat(fx_mul32x16(fx_neg_sat(coef[2 * n - 1 - j]), win[2 * n - 2 * j - 1]), fx_mul32x16(prev[j], win[2 * n - 2 * j - 2]));
  }
  return fx_shr_rnd(a, 16 - (q + 1));
}

/* lpfuncs.c:94: long block beside a short edge (start: the edge is on the left) */
FX_HD void x9_win_edge(const int32_t *y, const int32_t *ov, const X9Sink &sk, const int16_t *wl, const int16_t *ws, int q,
                       bool start, int lane, int nl) {
  constexpr int u = X9_U;
  if (start) {
    X9_FOR(i, 7 * u) {
      int32_t t = fx_shl_dir_sat_limit(fx_mul32x16(y[8 * u + i], wl[2 * i]), q + 1);
      sk.put(i, fx_add_sat(t, fx_shlw(ov[i], 16)));
      t = fx_shl_dir_sat_limit(fx_mul32x16(fx_neg(y[15 * u - 1 - i]), wl[2 * (7 * u - i) - 1]), q);
      sk.put(i + 9 * u, fx_shlw(t, 1));
    }
  } else {
    X9_FOR(i, 7 * u) {
      sk.put(i, fx_mul32x16_nosh_sat(ov[8 * u - 1 - i], fx_neg16(wl[2 * i + 1])));
      sk.put(9 * u + i, fx_sub_sat(fx_shl_dir_sat_limit(fx_neg(y[15 * u - 1 - i]), q - 1),
                                   fx_mul32x16_nosh_sat(ov[i + u], wl[14 * u - 2 - 2 * i])));
    }
  }
  const int16_t *wa = start ? wl + 14 * u : ws;
  const int16_t *wb = start ? ws : wl + 14 * u;
  X9_FOR(i, u) {
    const int32_t c = y[15 * u + i];
    const int32_t p = start ? ov[8 * u - 1 - i] : ov[u - 1 - i];
    const int16_t w1 = wa[2 * i], w2 = wa[2 * i + 1], w4 = wb[2 * i], w3 = wb[2 * i + 1];
    const int32_t a = fx_sub_sat(fx_shl_dir_sat_limit(fx_mul32x16(c, w1), q), fx_mul32x16_nosh_sat(p, w3));
    const int32_t b = fx_sub_sat(fx_shl_dir_sat_limit(fx_mul32x16(fx_neg_sat(c), w2), q), fx_mul32x16_nosh_sat(p, w4));
    sk.put(7 * u + i, fx_shlw(a, start ? 1 : 0));
    sk.put(9 * u - 1 - i, fx_shlw(b, start ? 1 : 0));
  }
}

/* lpfuncs.c:180-284: EIGHT_SHORT after a long-tailed frame; also the new overlap[0 .. u) */
FX_HD void x9_short_after_long(const int32_t *y, const int32_t *ov, const X9Sink &sk, int32_t *ovl_out, const int16_t *wsc,
                               const int16_t *wsp, const int16_t *wlp, int q, int lane, int nl) {
  constexpr int u = X9_U;
  X9_FOR(i, 7 * u) sk.put(i, fx_mul32x16_nosh_sat(ov[8 * u - 1 - i], fx_neg16(wlp[2 * i + 1])));
  X9_FOR(i, u) {
    sk.put(7 * u + i, fx_sub_sat(fx_shl_dir_sat_limit(fx_mul32x16(y[u + i], wsp[2 * i]), q),
                                 fx_mul32x16_nosh_sat(ov[u - 1 - i], wlp[14 * u + 1 + 2 * i])));
    sk.put(8 * u + i, fx_sub_sat(fx_shl_dir_sat_limit(fx_mul32x16(fx_neg_sat(y[2 * u - 1 - i]), wsp[2 * u - 2 * i - 1]), q),
                                 fx_mul32x16_nosh_sat(ov[i], wlp[16 * u - 2 - 2 * i])));
  }
  for (int b = 0; b < 4; b++) X9_FOR(i, u) {
    const int inc = 2 * u * b;
    const int32_t *cur = y + u + inc;
    const int32_t *pv = ov + u + inc;
    const int16_t *wl = wlp + 2 * (7 * u - inc);
    const int32_t c1 = cur[2 * u + i], c2 = cur[-1 - i];
    const int16_t sh1 = wsc[2 * i + 1], sh2 = wsc[2 * i];
    const int32_t a = fx_sub(fx_mul32x16(c1, sh2), fx_mul32x16(c2, sh1));
    sk.put(9 * u + inc + i, fx_sub_sat(fx_shl_dir_sat_limit(a, q), fx_mul32x16_nosh_sat(pv[i], wl[-2 - 2 * i])));
    if (b != 3) {
      const int32_t d = fx_sub(fx_mul32x16(fx_neg_sat(c1), sh1), fx_mul32x16(c2, sh2));
      sk.put(9 * u + inc + 2 * u - 1 - i,
             fx_sub_sat(fx_shl_dir_sat_limit(d, q), fx_mul32x16_nosh_sat(pv[2 * u - 1 - i], wl[-4 * u + 2 * i])));
    }
  }
  X9_FOR(i, u) {
    const int32_t a = fx_sub(fx_mul32x16(fx_neg(y[10 * u - 1 - i]), wsc[2 * u - 2 * i - 1]), fx_mul32x16(y[6 * u + i], wsc[2 * u - 2 * i - 2]));
    ovl_out[i] = fx_round16(fx_shl_dir_sat_limit(a, q + 1));
  }
}

/* block.c:1131 */
FX_HD int32_t x9_copy_outsample(int32_t ov) { return fx_shlw((int32_t)fx_sat16(fx_shlw((int32_t)(int16_t)ov, 1)), 14); }

/* qshift_adj as ixheaacd_imdct_process leaves it (lpfuncs.c:436-655, :757-775) */
FX_HD int x9_qshift_adj(int seq, bool prev_short_edge) {
  return ((seq == X9_ONLY_LONG || seq == X9_LONG_START) && prev_short_edge) ? 1 : 2;
}

/* The frame: spec[960] -> samples through sk, new overlap into ovl_new[480] (may be the place ov_old was loaded from: ov_old
   is a private copy).  y, a: 960 words of work space each.  headroom = norm32 over the frame's lines (aac_tns.c:422). */
FX_HD void x9_imdct_process(const int32_t *spec, const int32_t *ov_old, int32_t *ovl_new, int32_t *y, int32_t *a, int headroom,
                            int seq, int shape, int pseq, int pshape, const X9Sink &sk, int lane, int nl, bool ov_into_a = false) {
  constexpr int u = X9_U;
  const bool prev_short_edge = pseq == X9_LONG_START || pseq == X9_EIGHT_SHORT;
  const int16_t *wl = x9_long_win(pshape), *ws = x9_short_win(pshape);
  const int e = headroom - 1;
  if (seq != X9_EIGHT_SHORT) {
    const int q = x9_long_transform(spec, y, a, e, lane, nl);
    if (ov_into_a) { /* the work array is free now: the old overlap moves there before the new one overwrites its source */
      X9_FOR(i, 8 * u) a[i] = ov_old[i];
      x9_sync();
      ov_old = a;
    }
    if (seq == X9_ONLY_LONG) {
      if (!prev_short_edge) {
        x9_ola1(y, ov_old, sk, 0, wl, q, 8 * u, lane, nl); /* lpfuncs.c:444-452 */
      } else {
        x9_win_edge(y, ov_old, sk, wl, ws, q, true, lane, nl);
      }
      X9_FOR(i, 8 * u) ovl_new[i] = x9_to_ovl(y[i], q);
    } else if (seq == X9_LONG_START) {
      if (!prev_short_edge) {
        x9_ola1(y, ov_old, sk, 0, wl, q, 8 * u, lane, nl);
      } else {
        x9_win_edge(y, ov_old, sk, wl, ws, q, true, lane, nl);
      }
      X9_FOR(i, 7 * u) ovl_new[i] = fx_shr_rnd(fx_neg_sat(y[8 * u - 1 - i]), 16 - q); /* lpfuncs.c:286 */
      X9_FOR(i, u) ovl_new[7 * u + i] = x9_to_ovl(y[i], q);
    } else { /* LONG_STOP */
      if (prev_short_edge) {
        X9_FOR(i, 7 * u) sk.put(i, x9_copy_outsample(ov_old[i]));
        x9_ola1(y + 14 * u, ov_old + 7 * u, sk, 7 * u, ws, q, u, lane, nl);
        X9_FOR(i, 7 * u) sk.put(9 * u + i, fx_shl_dir_sat_limit(fx_neg_sat(y[15 * u - 1 - i]), q - 1)); /* :297 */
      } else {
        x9_win_edge(y, ov_old, sk, wl, ws, q, false, lane, nl);
      }
      X9_FOR(i, 8 * u) ovl_new[i] = x9_to_ovl(y[i], q);
    }
  } else {
    const int16_t *wsc = x9_short_win(shape);
    const int q = x9_short_transform(spec, y, a, e, lane, nl);
    if (ov_into_a) {
      X9_FOR(i, 8 * u) a[i] = ov_old[i];
      x9_sync();
      ov_old = a;
    }
    if (prev_short_edge) {
      X9_FOR(i, 7 * u) sk.put(i, fx_shl_sat((int32_t)(int16_t)ov_old[i], 15)); /* lpfuncs.c:325 */
      x9_ola1(y, ov_old + 7 * u, sk, 7 * u, ws, q, u, lane, nl);
      for (int b = 0; b < 3; b++) X9_FOR(i, u) { /* ola1 against the (requantised) tail of the previous short window */
        const int32_t *coef = y + 2 * u + 2 * u * b;
        const int16_t w1 = wsc[2 * u - 2 * i - 1], w2 = wsc[2 * u - 2 * i - 2];
        const int32_t c = coef[2 * u - 1 - i], pr = x9_to_ovl(y[2 * u * b + i], q);
        sk.put(9 * u + 2 * u * b + u - 1 - i, fx_sub_sat(fx_shl_dir_sat_limit(fx_mul32x16(c, w2), q), fx_mul32x16_nosh_sat(pr, w1)));
        sk.put(9 * u + 2 * u * b + u + i, fx_sub_sat(fx_shl_dir_sat_limit(fx_mul32x16(fx_neg_sat(c), w1), q), fx_mul32x16_nosh_sat(pr, w2)));
      }
      X9_FOR(i, u) {
        const int32_t t_lo = x9_ola2_value(y + 8 * u, y + 6 * u, wsc, q, u, i);
        const int32_t t_hi = x9_ola2_value(y + 8 * u, y + 6 * u, wsc, q, u, u + i);
        sk.put(15 * u + i, fx_shl_sat((int32_t)(int16_t)t_lo, 15)); /* lpfuncs.c:335 */
        ovl_new[i] = t_hi;
      }
    } else {
      x9_short_after_long(y, ov_old, sk, ovl_new, wsc, ws, wl, q, lane, nl);
    }
    for (int b = 0; b < 3; b++)
      X9_FOR(i, 2 * u) ovl_new[u + 2 * u * b + i] = x9_ola2_value(y + 10 * u + 2 * u * b, y + 8 * u + 2 * u * b, wsc, q, u, i);
    X9_FOR(i, u) ovl_new[7 * u + i] = x9_to_ovl(y[14 * u + i], q);
  }
}

#endif /* XAAC_IMDCT960_H */
