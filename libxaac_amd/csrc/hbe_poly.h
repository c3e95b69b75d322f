/*
 * hbe_poly.h -- the two polyphase banks of the QMF-domain harmonic transposer as host/device code: the real-valued
 * synthesis bank that turns the core band's QMF columns back into a sub-sampled time signal
 * (ixheaacd_real_synth_filt, decoder/ixheaacd_esbr_polyphase.c:157-274) and the complex analysis bank of twice the
 * size that follows it (ixheaacd_complex_anal_filt, :48-155), with the float FFTs they call
 * (common/ixheaac_esbr_fft.c).  Included by the oracle (oracle/oracle_hbe.cpp, sequential) and by the kernels
 * (hbe_kernel.hip); both are compiled without floating-point contraction, every expression keeps the reference's
 * operand order and width (FLOAT32 throughout).
 *
 * Both banks are written column by column in the reference (a delay line shifted per column).  Here a column's work is
 * a pure function of the frame's input and of the previous frame's delay line, so the columns of a frame are
 * independent:
 *   synthesis  column idx: v[idx][0 .. 2S) = transform of the column's S modulated inputs (xh_synth_column);
 *              output sample i = sum over the ten window blocks j of v[idx - j][(j odd ? S : 0) + i] * window[S j + i]
 *              in the order j = 0..9 (xh_synth_out); v of the nine columns before the frame is the delay line.
 *   analysis   column idx: u[i] = sum over five window blocks of the time signal (xh_anal_u), then the modulation
 *              transform (xh_anal_column).
 */
#ifndef XAAC_HBE_POLY_H
#define XAAC_HBE_POLY_H

#include <stdint.h>

#include "../../include/xaac_hbe.h"
#include "fx.h"

#if defined(__HIPCC__)
#define XAAC_TAB_QUAL static __device__ const
#include "tables_hbe.inc"
#undef XAAC_TAB_QUAL
#else
#include "tables_hbe.inc"
#endif

#pragma clang fp contract(off)

/* ---- tables per bank size (hbe_trans.c:71-100, :127-170) -------------------------------------------------------- */
FX_HD bool xh_size_ok(int s) { return s == 4 || s == 8 || s == 12 || s == 16 || s == 20; }
FX_HD const float *xh_window(int len) { /* ixheaacd_map_prot_filter */
  switch (len) {
    case 8: return xaac_hbe_window + 40;
    case 12: return xaac_hbe_window + 120;
    case 16: return xaac_hbe_window + 240;
    case 20: return xaac_hbe_window + 400;
    case 24: return xaac_hbe_window + 600;
    case 32: return xaac_hbe_window + 840;
    case 40: return xaac_hbe_window + 1160;
    default: return xaac_hbe_window;
  }
}
FX_HD const float *xh_synth_cos(int s) {
  switch (s) {
    case 8: return xaac_hbe_synth_cos_8;
    case 12: return xaac_hbe_synth_cos_12;
    case 16: return xaac_hbe_synth_cos_16;
    case 20: return xaac_hbe_synth_cos_20;
    default: return xaac_hbe_synth_cos_4;
  }
}
FX_HD const float *xh_analy_cs(int s) { /* the analysis bank has 2 s channels */
  switch (s) {
    case 8: return xaac_hbe_analy_cs_16;
    case 12: return xaac_hbe_analy_cs_24;
    case 16: return xaac_hbe_analy_cs_32;
    case 20: return xaac_hbe_analy_cs_40;
    default: return xaac_hbe_analy_cs_8;
  }
}

/* ---- the float FFTs (common/ixheaac_esbr_fft.c) ------------------------------------------------------------------ */
/* tw[j] = cos, tw[j + 257] = sin of 2 pi j / 1024 (ixheaac_twiddle_table_fft_float) */
FX_HD void xh_rot_a(float &r, float &i, float c, float s) { /* esbr_fft.c:186-188 */
  const float t = (r * c) + (i * s);
  i = -(r * s) + i * c;
  r = t;
}
FX_HD void xh_rot_b(float &r, float &i, float w_lo, float w_hi) { /* :284-286: (w_hi, w_lo) = table entries [.. + 1], [.. - 256] */
  const float t = (r * w_hi) - (i * w_lo);
  i = (r * w_lo) + (i * w_hi);
  r = t;
}
FX_HD void xh_rot_c(float &r, float &i, float c, float s) { /* :435-437 */
  const float t = -(r * c) - (i * s);
  i = -(r * s) + i * c;
  r = t;
}
/* the radix-4 butterfly of every pass (esbr_fft.c:122-138; alt: the last twiddle quadrant's form, :446-449).  v = x0r,
   x0i, x1r, x1i, x2r, x2i, x3r, x3i on entry, the reference's store order on return: x0, x2, x1, (x3i, x3r). */
FX_HD void xh_bfly4(float *v, bool alt) {
  float x0r = v[0], x0i = v[1], x1r = v[2], x1i = v[3], x2r = v[4], x2i = v[5], x3r = v[6], x3i = v[7];
  x0r = x0r + x2r;
  x0i = x0i + x2i;
  x2r = x0r - (x2r * 2);
  x2i = x0i - (x2i * 2);
  x1r = x1r + x3r;
  if (!alt) {
    x1i = x1i + x3i;
    x3r = x1r - (x3r * 2);
    x3i = x1i - (x3i * 2);
  } else {
    x1i = x1i - x3i;
    x3r = x1r - (x3r * 2);
    x3i = x1i + (x3i * 2);
  }
  x0r = x0r + x1r;
  x0i = x0i + x1i;
  x1r = x0r - (x1r * 2);
  x1i = x0i - (x1i * 2);
  x2r = x2r - x3i;
  x2i = x2i + x3r;
  x3i = x2r + (x3i * 2);
  x3r = x2i - (x3r * 2);
  v[0] = x0r; v[1] = x0i;
  v[2] = x2r; v[3] = x2i;
  v[4] = x1r; v[5] = x1i;
  v[6] = x3i; v[7] = x3r;
}
FX_HD unsigned xh_dig_rev(unsigned i, int m) { /* DIG_REV, esbr_fft.c:28-35 */
  unsigned v = i;
  v = ((v & 0x33333333u) << 2) | ((v & ~0x33333333u) >> 2);
  v = ((v & 0x0F0F0F0Fu) << 4) | ((v & ~0x0F0F0F0Fu) >> 4);
  v = ((v & 0x00FF00FFu) << 8) | ((v & ~0x00FF00FFu) >> 8);
  return v >> m;
}
FX_HD int xh_log2(int n) { return n == 8 ? 3 : (n == 16 ? 4 : (n == 32 ? 5 : 6)); } /* n = 8, 16, 32, 64 */

/* a "team" of one: the sequential run of the cooperative routines below (the oracle, and single columns) */
struct XhSeq {
  int lane, n;
  FX_MEMBER void sync() const {}
};

/* ixheaac_real_synth_fft_p2 (:42) / ixheaac_cmplx_anal_fft_p2 (:537) for n = 8, 16, 32 or 64 points, `units` transforms
   side by side on a team of lanes (CX: lane, n, sync(); a pass's butterflies touch disjoint elements, so spreading
   them over lanes changes no operation): unit t's input through `in` (real: in(t, e) = sample e, e < n, the upper half
   zero; complex: in(t, 2 c) / in(t, 2 c + 1) = real / imaginary part of point c), output y + t * ys (2 n floats). */
template <class CX, class IN>
FX_HD void xh_fft_p2_team(const CX &cx, const IN &in, float *y, int ys, int units, int n, bool real) {
  const int lg = xh_log2(n), rev_shift = 15 - lg; /* norm32(n) + 1 - 16 */
  const bool odd = (lg & 1) != 0;                 /* not a power of four: a radix-2 pass at the end */
  const int q4 = n / 4;
  for (int e = cx.lane; e < units * q4; e += cx.n) {
    const int t = e / q4, b = e % q4;
    unsigned h2 = xh_dig_rev((unsigned)(4 * b), rev_shift);
    if (odd) h2 = (h2 + 1) & ~1u;
    float *o = y + (size_t)t * ys + 8 * b;
    if (real) {
      const int i0 = (int)(h2 >> 1);
      float x0r = in(t, i0), x1r = in(t, i0 + (n >> 2)), x2r = in(t, i0 + 2 * (n >> 2)), x3r = in(t, i0 + 3 * (n >> 2));
      x0r = x0r + x2r;
      x2r = x0r - (x2r * 2);
      x1r = x1r + x3r;
      x3r = x1r - (x3r * 2);
      x0r = x0r + x1r;
      x1r = x0r - (x1r * 2);
      o[0] = x0r; o[1] = 0;
      o[2] = x2r; o[3] = x3r;
      o[4] = x1r; o[5] = 0;
      o[6] = x2r; o[7] = -x3r;
    } else {
      float v[8];
      for (int q = 0; q < 4; q++) {
        v[2 * q] = in(t, (int)h2 + q * (n >> 1));
        v[2 * q + 1] = in(t, (int)h2 + q * (n >> 1) + 1);
      }
      xh_bfly4(v, false);
      for (int q = 0; q < 8; q++) o[q] = v[q];
    }
  }
  cx.sync();
  const float *tw = xaac_hbe_fft_tw;
  int del = 4;
  for (int pass = (lg >> 1) - 1; pass > 0; pass--, del <<= 2) {
    for (int e = cx.lane; e < units * q4; e += cx.n) {
      const int t = e / q4, b = e % q4;
      float *yt = y + (size_t)t * ys;
      const int jj = b % del, k = b / del; /* twiddle column, group */
      const int p0 = 4 * del * k + jj;     /* complex index of the first leg; the others del apart */
      float v[8];
      for (int q = 0; q < 4; q++) {
        v[2 * q] = yt[2 * (p0 + q * del)];
        v[2 * q + 1] = yt[2 * (p0 + q * del) + 1];
      }
      bool alt = false;
      if (jj) {
        const int j = jj * (256 / del); /* nodespacing * jj; nodespacing * del = 256 in every pass */
        xh_rot_a(v[2], v[3], tw[j], tw[j + 257]);
        if (j <= 128) xh_rot_a(v[4], v[5], tw[2 * j], tw[2 * j + 257]);
        else xh_rot_b(v[4], v[5], tw[2 * j - 256], tw[2 * j + 1]);
        if (j <= 85) xh_rot_a(v[6], v[7], tw[3 * j], tw[3 * j + 257]);
        else if (j <= 170) xh_rot_b(v[6], v[7], tw[3 * j - 256], tw[3 * j + 1]);
        else {
          xh_rot_c(v[6], v[7], tw[3 * j - 512], tw[3 * j - 512 + 257]);
          alt = true;
        }
      }
      xh_bfly4(v, alt);
      for (int q = 0; q < 4; q++) {
        yt[2 * (p0 + q * del)] = v[2 * q];
        yt[2 * (p0 + q * del) + 1] = v[2 * q + 1];
      }
    }
    cx.sync();
  }
  if (odd) { /* :484-534: del = n / 2 */
    const int ns = 2 * (256 / del) * 1; /* nodespacing after the passes, doubled */
    for (int e = cx.lane; e < units * del; e += cx.n) {
      const int t = e / del, m = e % del;
      float *yt = y + (size_t)t * ys;
      const int tt = (m % (del / 2)) * ns;
      const float w1 = tw[tt], w4 = tw[tt + 257];
      const float x0r = yt[2 * m], x0i = yt[2 * m + 1];
      float x1r = yt[2 * (m + del)], x1i = yt[2 * (m + del) + 1];
      if (m < del / 2) {
        xh_rot_a(x1r, x1i, w1, w4);
      } else {
        const float tmp = (x1r * w4) - (x1i * w1);
        x1i = (x1r * w1) + (x1i * w4);
        x1r = tmp;
      }
      yt[2 * (m + del)] = x0r - x1r;
      yt[2 * (m + del) + 1] = x0i - x1i;
      yt[2 * m] = x0r + x1r;
      yt[2 * m + 1] = x0i + x1i;
    }
    cx.sync();
  }
}

FX_HD void xh_fft3(const float *inp, float *op) { /* ixheaac_aac_ld_dec_fft_3_float, esbr_fft.c:1048 */
  const float sinmu = -0.866025403784439f;
  const float temp_real = inp[0] + inp[2], temp_imag = inp[1] + inp[3];
  const float add_r = inp[2] + inp[4], add_i = inp[3] + inp[5];
  const float sub_r = inp[2] - inp[4], sub_i = inp[3] - inp[5];
  const float p1 = add_r / 2.0f, p4 = add_i / 2.0f, p2 = sub_i * sinmu, p3 = sub_r * sinmu;
  const float temp = inp[0] - p1;
  op[0] = temp_real + inp[4];
  op[1] = temp_imag + inp[5];
  op[2] = temp + p2;
  op[3] = (inp[1] - p3) - p4;
  op[4] = temp - p2;
  op[5] = (inp[1] + p3) - p4;
}

/* ixheaac_real_synth_fft_p3 (:1084, n = 24: 12 real samples + 12 zeros) and ixheaac_cmplx_anal_fft_p3 (:1148, n = 48
   complex points), `cols` of them side by side: three interleaved power-of-two transforms per column (3 cols units of
   xh_fft_p2_team into ysub, 2 n floats per column), then per (column, g) the two twiddles, the 3-point transform and the
   reference's output order -- the arrays the reference passes between these steps hold nothing another g reads.
   in as above (of the n-point transform); out + c * os: 2 n floats. */
template <class CX, class IN>
FX_HD void xh_fft_p3_team(const CX &cx, const IN &in, float *ysub, float *out, int os, int cols, int n, bool real) {
  const int m = n / 3; /* 8 or 16 points per sub-transform */
  const auto sub_in = [&](int t, int e) { /* unit t = 3 column + i: point j of it is point 3 j + i of the column */
    const int c = t / 3, i = t % 3;
    return real ? in(c, 3 * e + i) : in(c, 6 * (e >> 1) + 2 * i + (e & 1));
  };
  xh_fft_p2_team(cx, sub_in, ysub, 2 * m, 3 * cols, m, real);
  const float *wr = real ? xaac_hbe_tw24 : xaac_hbe_tw48;
  for (int e = cx.lane; e < cols * m; e += cx.n) {
    const int c = e / m, g = e % m;
    const float *ys = ysub + (size_t)c * 6 * m;
    float x[6], y[6];
    for (int q = 0; q < 3; q++) {
      x[2 * q] = ys[2 * m * q + 2 * g];
      x[2 * q + 1] = ys[2 * m * q + 2 * g + 1];
    }
    for (int q = 1; q < 3; q++) {
      const float cw = wr[4 * g + 2 * (q - 1)], sw = wr[4 * g + 2 * (q - 1) + 1];
      const float tmp = (x[2 * q] * cw + x[2 * q + 1] * sw);
      x[2 * q + 1] = (-x[2 * q] * sw + x[2 * q + 1] * cw);
      x[2 * q] = tmp;
    }
    xh_fft3(x, y);
    float *o = out + (size_t)c * os;
    for (int q = 0; q < 3; q++) {
      o[2 * m * q + 2 * g] = y[2 * q];
      o[2 * m * q + 2 * g + 1] = y[2 * q + 1];
    }
  }
  cx.sync();
}
#define XH_FFT_SCRATCH 512 /* floats the single-column wrappers below take as work space */
#define XH_SYNTH_SCRATCH 264

/* ---- the real synthesis bank ------------------------------------------------------------------------------------- */
/* `cols` columns side by side (esbr_polyphase.c:186-247): xin(c, k) = the column's k-th modulated input, k < s
   (xh_synth_xin); v(c) = where column c's 2 s values go (what the reference writes to the front of its delay line);
   work: cols * 128 floats (transform output, and the sub-transforms of the 24-point case). */
FX_HD float xh_synth_xin(const float *re, const float *im, int k_start, int k) { /* :166-168, :189-191 */
  const float *ct = xaac_hbe_cos_trans_qmf + k_start * 32;
  return (ct[2 * k] * re[k_start + k] + ct[2 * k + 1] * im[k_start + k]);
}
template <class CX, class XIN, class VOUT>
FX_HD void xh_synth_team(const CX &cx, const XIN &xin, const VOUT &v, int cols, int s, float *work) {
  const float *tab = xh_synth_cos(s);
  if (s == 20) {
    /* :199-221: 31 dot products.  l <= s writes entries l and s - l, l > s entries l and 3 s - l (negated).  In the
       reference's serial order the stores of l > s / 2 land on top of those of s - l, so of l = 0 .. s only l >= s / 2
       leave anything: v[s - l] and v[l] (once for l = s / 2).  One l per lane, the overwritten stores left out. */
    for (int e = cx.lane; e < cols * (3 * s / 2 + 1); e += cx.n) {
      const int c = e / (3 * s / 2 + 1), l = e % (3 * s / 2 + 1);
      float accu = 0.0f;
      for (int k = 0; k < s; k++) accu += xin(c, k) * tab[l * s + k];
      float *vc = v(c);
      if (l <= s) {
        if (l >= s / 2) {
          vc[l] = accu;
          vc[s - l] = accu;
        }
      } else if (l < 3 * s / 2) {
        vc[l] = accu;
        vc[3 * s - l] = -accu;
      } else {
        vc[3 * s / 2] = accu;
      }
    }
    cx.sync();
    return;
  }
  const int n = 2 * s;
  const auto in = [&](int c, int e) { return e < s ? xin(c, e) : 0.0f; }; /* :192 */
  float *u = work;
  int us = 2 * n;
  if (s == 12) {
    u = work + (size_t)cols * 2 * n;
    xh_fft_p3_team(cx, in, work, u, us, cols, n, true);
  } else {
    xh_fft_p2_team(cx, in, u, us, cols, n, true);
  }
  const int kmax = s / 2;
  for (int e = cx.lane; e < cols * n; e += cx.n) { /* :233-246: the first 3 s / 2 results go to v[s / 2 ..], the rest to v[0 ..] */
    const int c = e / n, k = e % n;
    const float *uc = u + (size_t)c * us;
    float tmp = (uc[2 * k] * tab[2 * k]);
    tmp -= (uc[2 * k + 1] * tab[2 * k + 1]);
    v(c)[k < kmax + s ? kmax + k : k - (kmax + s)] = tmp;
  }
  cx.sync();
}
/* One column: re / im = the column's 64 QMF bands; v[0 .. 2 s); w: XH_FFT_SCRATCH floats. */
FX_HD void xh_synth_column(const float *re, const float *im, int s, int k_start, float *v, float *w) {
  const XhSeq cx = {0, 1};
  xh_synth_team(cx, [&](int, int k) { return xh_synth_xin(re, im, k_start, k); }, [&](int) { return v; }, 1, s, w);
}
/* Output sample i of column idx (:249-268).  vv(c, t): element t of column c's v, c = -9 .. num_columns - 1 (the
   negative ones from the delay line: xh_synth_hist). */
template <class VV>
FX_HD float xh_synth_out(const VV &vv, int s, int idx, int i) {
  const float *win = xh_window(s);
  float accu = 0.0f;
  for (int j = 0; j < 10; j++) accu = accu + vv(idx - j, (j & 1) ? s + i : i) * win[s * j + i];
  return accu;
}
/* element t of column c < 0 in the delay line as the previous frame left it: block -1 - c */
FX_HD float xh_synth_hist(const float *synth_buf, int s, int c, int t) { return synth_buf[2 * s * (-1 - c) + t]; }

/* ---- the complex analysis bank ----------------------------------------------------------------------------------- */
/* Sample n of the delay line at column idx (esbr_polyphase.c:92-98): block m = n / A holds column idx - m's A new
   samples, reversed; columns before the frame come from analy_buf (block -1 - c of the previous frame). */
FX_HD float xh_anal_x(const float *input_buf, const float *analy_buf, int a, int idx, int n) {
  const int c = idx - n / a, i = n % a;
  return c >= 0 ? input_buf[(c + 1) * a - i] : analy_buf[(-1 - c) * a + i];
}
FX_HD float xh_anal_u_w(const float *input_buf, const float *analy_buf, int a, int idx, int i, const float *win) { /* :100-109 */
  float accu = 0.0f;
  for (int j = 0; j < 5; j++) {
    const int n = i + j * 2 * a;
    accu = accu + xh_anal_x(input_buf, analy_buf, a, idx, n) * win[n];
  }
  return accu;
}
FX_HD float xh_anal_u(const float *input_buf, const float *analy_buf, int a, int idx, int i) {
  return xh_anal_u_w(input_buf, analy_buf, a, idx, i, xh_window(a));
}

/* ---- the DFT transposer's analysis bank (ixheaacd_dft_hbe_cplx_anal_filt, esbr_polyphase.c:276-338) --------------- */
/* prototype filter as ixheaacd_hbe_dft_trans.c:69-108 maps it: sizes 28 and 36 have their own table, 44 is the last of
   the concatenated windows, every size without a case (48 .. 64) falls back to the start of the concatenation, which the bank then reads 10 L entries of */
FX_HD const float *xh_window_dft(int len) {
  switch (len) {
    case 28: return xaac_hbe_window_28_36;
    case 36: return xaac_hbe_window_28_36 + 280;
    case 44: return xaac_hbe_window + 1560;
    default: return xh_window(len);
  }
}
/* sub-band k of a column: the sums over 2 L windowed-and-folded samples u[] against row k of the transposer's coefficient
   matrices ([64][128], made by ixheaacd_dft_hbe_data_reinit, hbe_dft_trans.c:374-388) */
FX_HD void xh_dft_anal_band(const float *u, int l2, const float *coef_re_row, const float *coef_im_row, float &out_r, float &out_i) {
  float accu_r = 0, accu_i = 0;
  for (int l = 0; l < l2; l++) {
    accu_r = accu_r + u[l] * coef_re_row[l];
    accu_i = accu_i + u[l] * coef_im_row[l];
  }
  out_r = accu_r;
  out_i = accu_i;
}
/* `cols` columns side by side (:110-152): u + c * us = the column's 2 a windowed sums (overwritten when a = 40);
   out(c) = where its a complex sub-band samples (2 a floats) go; work: cols * 256 floats (a = 40: none). */
template <class CX, class OUT>
FX_HD void xh_anal_team(const CX &cx, float *u, int us, const OUT &out, int cols, int a, float *work) {
  const float *tab = xh_analy_cs(a / 2);
  if (a == 40) {
    for (int e = cx.lane; e < cols * (a - 1); e += cx.n) {
      float *uc = u + (size_t)(e / (a - 1)) * us;
      const int i = 1 + e % (a - 1);
      const float t1 = uc[i] + uc[2 * a - i], t2 = uc[i] - uc[2 * a - i];
      uc[i] = t1;
      uc[2 * a - i] = t2;
    }
    cx.sync();
    for (int e = cx.lane; e < cols * a; e += cx.n) {
      const int c = e / a, k = e % a;
      const float *uc = u + (size_t)c * us;
      float accu_r = uc[a], accu_i = (k & 1) ? uc[0] : -uc[0];
      for (int l = 1; l < a; l++) {
        accu_r = accu_r + uc[l] * tab[2 * a * k + 2 * l];
        accu_i = accu_i + uc[2 * a - l] * tab[2 * a * k + 2 * l + 1];
      }
      float *o = out(c);
      o[2 * k] = accu_r;
      o[2 * k + 1] = accu_i;
    }
    cx.sync();
    return;
  }
  const int n = 2 * a;
  const auto in = [&](int c, int e) { return (tab[e] * u[(size_t)c * us + (e >> 1)]); }; /* u_in[2 k] , [2 k + 1] = tab * u[k] */
  float *y = work;
  const int ys = 2 * n;
  if (a == 24) {
    y = work + (size_t)cols * 2 * n;
    xh_fft_p3_team(cx, in, work, y, ys, cols, n, false);
  } else {
    xh_fft_p2_team(cx, in, y, ys, cols, n, false);
  }
  for (int e = cx.lane; e < cols * (a / 2); e += cx.n) {
    const int c = e / (a / 2), k = e % (a / 2);
    const float *yc = y + (size_t)c * ys;
    float *o = out(c);
    o[4 * k + 1] = -yc[4 * k];
    o[4 * k] = yc[4 * k + 1];
    o[4 * k + 3] = yc[4 * k + 2];
    o[4 * k + 2] = -yc[4 * k + 3];
  }
  cx.sync();
}
/* One column: u[0 .. 2 a) -> out[0 .. 2 a).  u is overwritten; w: XH_FFT_SCRATCH floats. */
FX_HD void xh_anal_column(float *u, int a, float *out, float *w) {
  const XhSeq cx = {0, 1};
  xh_anal_team(cx, u, 0, [&](int) { return out; }, 1, a, w);
}

#endif /* XAAC_HBE_POLY_H */
