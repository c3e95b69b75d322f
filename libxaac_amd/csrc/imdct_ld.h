/*
 * imdct_ld.h -- the 512- and 480-line inverse transforms of AAC-LD and AAC-ELD with their windowing / overlap-add, shared
 * by the gfx950 kernel (imdct_ld_kernel.hip) and, compiled for the host, by the checker (oracle/oracle_imdct_ld.cpp).
 *
 * Restates the frame_length 512 / 480 branches of ixheaacd_imdct_process (decoder/ixheaacd_lpfuncs.c:385-409, :456-486):
 *   ixheaacd_inverse_transform_512 / ixheaacd_mdct_480_ld                 decoder/ixheaacd_aac_imdct.c:1761 / :1707
 *   ixheaacd_pre_twiddle, ixheaacd_post_twiddle_ld / _eld                 aac_imdct.c:2577 / :2744 / :2786
 *   ixheaacd_fft32x32_ld_dec (256 and 16 points)                          aac_imdct.c:2894
 *   ixheaacd_fft_480_ld = 15 x 16 points + 16 x ixheaacd_fft_15_ld_dec    aac_imdct.c:2461 / :3162
 *   ixheaacd_lap1_512_480 (block.c:1140) + ixheaacd_spec_to_overlapbuf_dec (lpfuncs.c:316)      -- LD
 *   the sign / copy step lpfuncs.c:401-408 + ixheaacd_eld_dec_windowing (lpfuncs.c:804-1010)    -- ELD
 * Both object types have ONLY_LONG frames only and hand PCM16 over directly (qshift_adj = -2).
 *
 * Form as in imdct960.h: every stage a loop over independent items (X9_FOR), one "lane" on the host, 64 on a wave.
 * Arithmetic: wrapping sums and (a * b) >> 32 products in the twiddles, saturating adds inside the FFTs with the
 * split-product twiddle multiplications MPYHIRC / MPYLUHS, saturating window sums.
 */
#ifndef XAAC_IMDCT_LD_H
#define XAAC_IMDCT_LD_H

#include "imdct960.h"

#ifndef XAAC_LD_TABLES_INCLUDED
#define XAAC_LD_TABLES_INCLUDED
#if defined(__HIPCC__)
#define XAAC_TAB_QUAL static __device__ const
#include "tables_imdct_ld.inc"
#undef XAAC_TAB_QUAL
#else
#include "tables_imdct_ld.inc"
#endif
#endif

/* work space per channel-frame (words): a = 2 F (the ELD transform's 2 F outputs), b = F */
#define XL_A_WORDS 1024
#define XL_B_WORDS 512
#define XL_OV_WORDS 1536 /* ELD: 3 F words of overlap; LD: F / 2 */

/* the (cos, sin) of complex element m in the interleaved (c, c1, s, s1) tables (aac_imdct.c:2593-2596, :2757-2760) */
template <int F>
FX_HD void xl_cs(int m, int32_t &c, int32_t &s) {
  const int k = 4 * (m >> 1) + (m & 1);
  c = F == 512 ? xaac_ld_cos_1024[k] : xaac_ld_cos_960[k];
  s = F == 512 ? xaac_ld_cos_1024[k + 2] : xaac_ld_cos_960[k + 2];
}

/* aac_imdct.c:2577: complex element m of the F / 2 the FFT takes */
template <int F>
FX_HD X9Cx xl_pre_twiddle(const int32_t *data, int m, int neg_expo) {
  int32_t c, s;
  xl_cs<F>(m, c, s);
  const int32_t tr = data[2 * m], ti = data[F - 1 - 2 * m];
  const int32_t r = fx_neg(fx_add(fx_mul32(tr, c), fx_mul32(ti, s))), i = fx_sub(fx_mul32(tr, s), fx_mul32(ti, c));
  X9Cx v;
  v.r = neg_expo >= 0 ? fx_shr(r, neg_expo) : fx_shl(r, -neg_expo);
  v.i = neg_expo >= 0 ? fx_shr(i, neg_expo) : fx_shl(i, -neg_expo);
  return v;
}

/* MPYHIRC / MPYLUHS (aac_imdct.c:52-59): the twiddle's halves against the sample's halves */
FX_HD int32_t xl_mpyhirc(int32_t x, int32_t y) {
  const int32_t h = (int16_t)(x >> 16);
  return fx_add((h * (int32_t)(uint16_t)(y & 0xffff) + 0x4000) >> 15, fx_shlw(h * (int32_t)(int16_t)(y >> 16), 1));
}
FX_HD int32_t xl_mpyluhs(int32_t x, int32_t y) { return (int32_t)(uint16_t)(x & 0xffff) * (int32_t)(int16_t)(y >> 16); }
/* s * a + c * b and c * a - s * b as the butterflies spell them (aac_imdct.c:2993-3015) */
FX_HD int32_t xl_rot_sum(int32_t s, int32_t a, int32_t c, int32_t b) {
  const int32_t lo = fx_add(fx_add(xl_mpyluhs(s, a), xl_mpyluhs(c, b)), 0x8000) >> 16;
  return fx_add(fx_add(xl_mpyhirc(s, a), xl_mpyhirc(c, b)), fx_shlw(lo, 1));
}
FX_HD int32_t xl_rot_diff(int32_t c, int32_t a, int32_t s, int32_t b) {
  const int32_t lo = fx_add(fx_sub(xl_mpyluhs(c, a), xl_mpyluhs(s, b)), 0x8000) >> 16;
  return fx_add(fx_sub(xl_mpyhirc(c, a), xl_mpyhirc(s, b)), fx_shlw(lo, 1));
}

/* one radix-4 butterfly of a pass (aac_imdct.c:2949-3016): complex positions p, p + h, p + 2h, p + 3h; w -> its 6 twiddles */
FX_HD void xl_fft_bfly(int32_t *x, int p, int h, const int32_t *w) {
  const X9Cx a = x9_ld(x, p), b = x9_ld(x, p + h), c = x9_ld(x, p + 2 * h), d = x9_ld(x, p + 3 * h);
  const int32_t si10 = w[0], co10 = w[1], si20 = w[2], co20 = w[3], si30 = w[4], co30 = w[5];
  const X9Cx xh = x9_add(a, c), xl = x9_sub(a, c), xh2 = x9_add(b, d), xl2 = x9_sub(b, d);
  x9_st(x, p, x9_add(xh, xh2));
  const int32_t xt0 = fx_sub_sat(xh.r, xh2.r), yt0 = fx_sub_sat(xh.i, xh2.i);
  const int32_t xt1 = fx_add_sat(xl.r, xl2.i), yt2 = fx_add_sat(xl.i, xl2.r);
  const int32_t xt2 = fx_sub_sat(xl.r, xl2.i), yt1 = fx_sub_sat(xl.i, xl2.r);
  X9Cx v;
  v.r = xl_rot_sum(si10, yt1, co10, xt1);
  v.i = xl_rot_diff(co10, yt1, si10, xt1);
  x9_st(x, p + h, v);
  v.r = xl_rot_sum(si20, yt0, co20, xt0);
  v.i = xl_rot_diff(co20, yt0, si20, xt0);
  x9_st(x, p + 2 * h, v);
  v.r = xl_rot_sum(si30, yt2, co30, xt2);
  v.i = xl_rot_diff(co30, yt2, si30, xt2);
  x9_st(x, p + 3 * h, v);
}

/* DIG_REV (aac_imdct.c:42): the 2-bit digits of j in reverse order, the top ones kept */
FX_HD int xl_dig_rev(unsigned j, int shift) {
  j = ((j & 0x33333333u) << 2) | ((j & ~0x33333333u) >> 2);
  j = ((j & 0x0F0F0F0Fu) << 4) | ((j & ~0x0F0F0F0Fu) >> 4);
  j = ((j & 0x00FF00FFu) << 8) | ((j & ~0x00FF00FFu) >> 8);
  j = ((j & 0x0000FFFFu) << 16) | ((j & ~0x0000FFFFu) >> 16);
  return (int)(j >> shift);
}

/* item `it` (0 .. N/4 - 1) of the last, digit-reversing radix-4 pass of an N-point transform (aac_imdct.c:3040-3147):
   four consecutive inputs -> outputs N / 4 apart */
template <int N>
FX_HD void xl_fft_last(const int32_t *x, int32_t *y, int it) {
  const int half = it & 1, k = it >> 1; /* k-th iteration of the reference's loop, its x0 (half 0) or x2 (half 1) block */
  const int lower = k < N / 16 ? 1 : 0;
  const int j = lower ? 4 * k : N / 2 + 4 * (k - N / 16);
  const int src = (lower ? 4 * k : N / 2 + 4 * (k - N / 16)) + half * (N / 4);
  const int h2 = xl_dig_rev((unsigned)j, N == 256 ? 24 : 28) + half;
  const X9Cx x0 = x9_ld(x, src), x1 = x9_ld(x, src + 1), x2 = x9_ld(x, src + 2), x3 = x9_ld(x, src + 3);
  const X9Cx xh0 = x9_add(x0, x2), xl0 = x9_sub(x0, x2), xh1 = x9_add(x1, x3), xl1 = x9_sub(x1, x3);
  X9Cx n1, n3;
  n1.r = fx_add_sat(xl0.r, xl1.i);
  n1.i = fx_sub_sat(xl0.i, xl1.r);
  n3.r = fx_sub_sat(xl0.r, xl1.i);
  n3.i = fx_add_sat(xl0.i, xl1.r);
  x9_st(y, h2, x9_add(xh0, xh1));
  x9_st(y, N / 4 + h2, n1);
  x9_st(y, N / 2 + h2, x9_sub(xh0, xh1));
  x9_st(y, 3 * N / 4 + h2, n3);
}

FX_HD int32_t xl_scale480(int32_t v) { return fx_mul32_shl(v, 1145324612); } /* aac_imdct.c:1713, :1738-1754 */

/* post twiddle of element j: LD (aac_imdct.c:2744) two outputs; ELD (:2786) four of the 2 F outputs, the copies and sign
   changes included; 480 lines: each stored value through the 1145324612 scale afterwards, as the reference's loop does */
template <int F, bool ELD>
FX_HD void xl_post_twiddle(X9Cx x, int j, int32_t *out) {
  int32_t c, s;
  xl_cs<F>(j, c, s);
  const int32_t ti = fx_sub(fx_mul32(x.i, c), fx_mul32(x.r, s)), tr = fx_neg(fx_add(fx_mul32(x.r, c), fx_mul32(x.i, s)));
#define XL_S(v) (F == 480 ? xl_scale480(v) : (v))
  if (!ELD) {
    out[F - 1 - 2 * j] = XL_S(ti);
    out[2 * j] = XL_S(tr);
  } else {
    constexpr int m = F;
    out[m + m / 2 - 1 - 2 * j] = XL_S(tr);
    out[m / 2 + 2 * j] = XL_S(ti);
    if (j < m / 4) {
      out[m + m / 2 + 2 * j] = XL_S(tr);
      out[m / 2 - 1 - 2 * j] = XL_S(fx_neg(ti));
    } else {
      const int t = j - m / 4;
      out[2 * t] = XL_S(fx_neg(tr));
      out[2 * m - 1 - 2 * t] = XL_S(ti);
    }
  }
#undef XL_S
}

/* spec[F] -> a (LD: F time-aliased values; ELD: the 2 F values the reference leaves at data + F); b: F words of work space.
   Returns q_shift.  e = headroom(spec) - 1. */
template <int F, bool ELD>
FX_HD int xl_transform(const int32_t *spec, int32_t *a, int32_t *b, int e, int lane, int nl) {
  const int neg_expo = 7 - e;
  if (F == 512) {
    X9_FOR(m, 256) x9_st(b, m, xl_pre_twiddle<512>(spec, m, neg_expo));
    x9_sync();
    X9_FOR(t, 64) xl_fft_bfly(b, t, 64, xaac_ld_w_256 + 6 * t);
    x9_sync();
    X9_FOR(t, 64) xl_fft_bfly(b, 64 * (t >> 4) + (t & 15), 16, xaac_ld_w_256 + 384 + 6 * (t & 15));
    x9_sync();
    X9_FOR(t, 64) xl_fft_bfly(b, 16 * (t >> 2) + (t & 3), 4, xaac_ld_w_256 + 480 + 6 * (t & 3));
    x9_sync();
    if (!ELD) { /* LD writes F words: the transform's outputs go to the upper half of a, the post twiddle fills the lower */
      X9_FOR(t, 64) xl_fft_last<256>(b, a + 512, t);
      x9_sync();
      X9_FOR(j, 256) xl_post_twiddle<512, ELD>(x9_ld(a + 512, j), j, a);
    } else { /* ELD writes all 2 F words of a: the 256 transform outputs pass through b */
      X9_FOR(t, 64) xl_fft_last<256>(b, a, t);
      x9_sync();
      X9_FOR(j, 256) {
        const X9Cx x = x9_ld(a, j);
        b[2 * j] = x.r;
        b[2 * j + 1] = x.i;
      }
      x9_sync();
      X9_FOR(j, 256) xl_post_twiddle<512, ELD>(x9_ld(b, j), j, a);
    }
    x9_sync();
    return 15 - e;
  } else {
    /* pre twiddle through re_arr_tab_16 (aac_imdct.c:2467) */
    X9_FOR(n, 240) x9_st(b, n, xl_pre_twiddle<480>(spec, xaac_ld_arr_16[n], neg_expo));
    x9_sync();
    X9_FOR(t, 60) xl_fft_bfly(b, 16 * (t >> 2) + (t & 3), 4, xaac_ld_w_16 + 6 * (t & 3));
    x9_sync();
    X9_FOR(t, 60) xl_fft_last<16>(b + 2 * 16 * (t >> 2), a + 2 * 16 * (t >> 2), t & 3);
    x9_sync();
    /* 16 x 15 points (aac_imdct.c:3162): 5-point stage on inputs 16 complex apart in the reference's order ... */
    X9_FOR(t, 48) {
      const int j = t & 15, g = t >> 4;
      X9Cx in[5], out[5];
#pragma unroll
      for (int m = 0; m < 5; m++) {
        const int k = 5 * g + 3 * m;
        in[m] = x9_ld(a, j + 16 * (k >= 15 ? k - 15 : k));
      }
      x9_fft5<true>(in, out);
#pragma unroll
      for (int m = 0; m < 5; m++) x9_st(b, 15 * j + 5 * g + m, out[m]);
    }
    x9_sync();
    /* ... 3-point stage, results through re_arr_tab_sml_240 */
    X9_FOR(t, 80) {
      const int j = t & 15, i = t >> 4;
      X9Cx out[3];
      x9_fft3(x9_ld(b, 15 * j + i), x9_ld(b, 15 * j + 5 + i), x9_ld(b, 15 * j + 10 + i), out);
#pragma unroll
      for (int m = 0; m < 3; m++) x9_st(a, xaac_ld_arr_sml_240[15 * j + 3 * i + m], out[m]);
    }
    x9_sync();
    X9_FOR(j, 240) {
      const X9Cx x = x9_ld(a, j);
      b[2 * j] = x.r;
      b[2 * j + 1] = x.i;
    }
    x9_sync();
    X9_FOR(j, 240) xl_post_twiddle<480, ELD>(x9_ld(b, j), j, a);
    x9_sync();
    return 16 - e;
  }
}

/* ---- LD: ixheaacd_lap1_512_480 (block.c:1140) and the new overlap (lpfuncs.c:460-470) ------------------------------- */
template <int F>
FX_HD void xl_ld_overlap_add(const int32_t *y, const int32_t *ov_old, int32_t *ov_new, int16_t *pcm, int stride, int q,
                             int shape_prev, int lane, int nl) {
  constexpr int size = F / 2;
  const int32_t *win = F == 512 ? (shape_prev ? xaac_ld_win_low_512 : xaac_ld_win_sine_512)
                                : (shape_prev ? xaac_ld_win_low_480 : xaac_ld_win_sine_480);
  X9_FOR(i, size) {
    const int16_t prev = (int16_t)ov_old[i];
    const int32_t win1 = win[size - 1 - i], win2 = win[size + i], coeff = y[2 * size - 1 - i];
    int32_t accu = fx_sub_sat(fx_shl_dir_sat_limit(fx_mul32_shl(coeff, win1), q), fx_add(-0x2000, fx_mul32x16_shl(win2, prev)));
    accu = fx_add_sat(accu, accu);
    accu = fx_add_sat(accu, accu);
    pcm[stride * (size - 1 - i)] = (int16_t)(accu >> 16);
    accu = fx_sub_sat(fx_shl_dir_sat_limit(fx_mul32_shl(fx_neg_sat(coeff), win2), q), fx_add(-0x2000, fx_mul32x16_shl(win1, prev)));
    accu = fx_add_sat(accu, accu);
    accu = fx_add_sat(accu, accu);
    pcm[stride * (size + i)] = (int16_t)(accu >> 16);
    ov_new[i] = fx_shr_rnd(y[i], 16 - q);
  }
}

/* ---- ELD: z = the 4 F values of lpfuncs.c:401-408 read out of the transform's 2 F; ixheaacd_eld_dec_windowing ---------- */
template <int F>
FX_HD int32_t xl_eld_z(const int32_t *out, int i) {
  return i < F ? fx_neg(out[F + i]) : i < 3 * F ? out[i - F] : fx_neg(out[i - 3 * F]);
}

template <int F>
FX_HD void xl_eld_overlap_add(const int32_t *out, const int32_t *ov_old, int32_t *ov_new, int16_t *pcm, int stride, int q_shift,
                              int lane, int nl) {
  constexpr int delay = F / 4;
  const int16_t *win = F == 512 ? xaac_ld_win_eld_512 : xaac_ld_win_eld_480;
  const int q = q_shift + 2;
  X9_FOR(n, F) {
    const int32_t w = fx_mul32x16(xl_eld_z<F>(out, delay + n), win[delay + n]);
    const int32_t v = fx_add_sat(q >= 0 ? fx_shl(w, q) : fx_shr(w, -q), ov_old[n]);
    pcm[stride * n] = fx_round16(q >= 0 ? fx_shl_sat(v, 1) : fx_shl(v, 1));
  }
  X9_FOR(k, 2 * F + F - delay) {
    const int32_t w = fx_mul32x16(xl_eld_z<F>(out, delay + F + k), win[delay + F + k]);
    const int32_t sh = q >= 0 ? fx_shl(w, q) : fx_shr(w, -q);
    ov_new[k] = k < 2 * F ? fx_add_sat(sh, ov_old[F + k]) : sh;
  }
}

#endif /* XAAC_IMDCT_LD_H */
