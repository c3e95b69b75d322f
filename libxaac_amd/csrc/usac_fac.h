/*
 * usac_fac.h -- the forward-aliasing-cancellation signal of a USAC FD frame behind an LPD frame, shared by the gfx950 kernel
 * (usac_imdct_kernel.hip: xaac_usac_fac_kernel) and, compiled for the host, by the checker (oracle/oracle_usac.cpp).
 *
 * Restates:
 *   ixheaacd_cal_fac_data                          decoder/ixheaacd_imdct.c:210
 *   ixheaacd_fr_alias_cnx_fix, _weighted_synthesis_filter, _synthesis_tool      ixheaacd_fwd_alias_cnx.c:138 / :60 / :72
 *   ixheaacd_acelp_mdct, _pre_twid, _post_twid     ixheaacd_acelp_mdct.c:166 / :102 / :129
 *   ixheaacd_complex_fft_p2_dec and _p3 with fft_mode = -1 (the forward transforms)   ixheaacd_fft.c:1449-1965 / :2531
 * The function is serial in its heart (an order-16 recursive filter over 2 lfac samples behind a 24- to 64-point transform), and it
 * only runs for frames behind an LPD frame: the code below is the reference's own order of operations, element-wise loops spread over
 * the lanes of a team (cx), the transform and the filter on one lane.  Float steps are single IEEE operations in the reference's
 * order (no contraction: the products feed conversions, not sums).
 */
#ifndef XAAC_USAC_FAC_H
#define XAAC_USAC_FAC_H

#include <string.h>

#include "usac_imdct.h"

struct XfCx {
  int lane, n;
  FX_MEMBER void sync() const {
#if defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
#endif
  }
};
#define XF_PAR(i, lo, hi) for (int i = (lo) + cx.lane; i < (hi); i += cx.n)
#define XF_ONE if (cx.lane == 0)

struct XfWork {
  int32_t x_in[128];          /* the re-ordered coefficients (imdct.c:283-286) */
  int32_t re[64], im[64];     /* the transform's points */
  int32_t y[128];             /* its work array */
  int32_t izir[264];          /* the zero-input response, zero behind n_long / 4 (the reference's array is 512 zeros) */
  int32_t aq[17], ap[17];     /* the LPC filter and its weighted form */
  int32_t out[16 + 256];      /* fac_idata: 16 zeros in front of the 2 lfac samples */
  int32_t scaled[129];        /* fac_data after its in-place scaling (:278-281) */
  float red[64];              /* lane maxima */
  int32_t ired[64];
  int32_t s_gain_fac, s_scale, s_q1, s_q2, s_q3, s_preshift, s_err, s_itemp, s_q_out;
  float s_qfac1;
};

/* (a * b) >> shift, ixheaac_mul32_sh (basic_ops40.h:235); the count is taken modulo 64 as the x86-64 build's shift does */
FX_HD int32_t xf_mul32_sh(int32_t a, int32_t b, int shift) { return (int32_t)(((int64_t)a * (int64_t)b) >> (shift & 63)); }
FX_HD int32_t xf_mult32_m(int32_t a, int32_t b) { return (int32_t)(((int64_t)a * (int64_t)b) >> 31); } /* fwd_alias_cnx.c:50 */
FX_HD int32_t xf_sat64(int64_t v) { return v >= 2147483647LL ? 2147483647 : (v <= -2147483648LL ? (int32_t)(-2147483647 - 1) : (int32_t)v); }
/* acelp_mdct.c:65 / :83: (a c -+ b d) >> 32, clamped */
FX_HD int32_t xf_mul_sub64(int32_t a, int32_t b, int32_t c, int32_t d) { return xf_sat64(((int64_t)a * c - (int64_t)b * d) >> 32); }
FX_HD int32_t xf_mul_add64(int32_t a, int32_t b, int32_t c, int32_t d) { return xf_sat64(((int64_t)a * c + (int64_t)b * d) >> 32); }
FX_HD float xf_bits(int32_t b) {
  float f;
  memcpy(&f, &b, 4);
  return f;
}
FX_HD float xf_pow2(int e) { return xf_bits((int32_t)((127 + e) << 23)); } /* (FLOAT32)((WORD64)1 << e), e = 0 .. 62 */

/* ---- ixheaacd_complex_fft_p2_dec, fft_mode = -1 (fft.c:1449-1965): npoints = 4 .. 512, a power of two; xr / xi in place;
   y: 2 npoints words; returns the exponent the reference reports through *preshift (with *preshift = 0 on entry) ------------- */
FX_HD void xf_rot_a(int32_t &r, int32_t &i, int32_t h, int32_t l) { /* :1595-1598 */
  const int32_t t = fx_sub_sat(xu_mul_sat(r, l), xu_mul_sat(i, h));
  i = fx_add_sat(xu_mul_sat(r, h), xu_mul_sat(i, l));
  r = t;
}
FX_HD void xf_rot_b(int32_t &r, int32_t &i, int32_t h, int32_t l) { /* :1683-1687 */
  const int32_t t = fx_add_sat(xu_mul_sat(r, h), xu_mul_sat(i, l));
  i = fx_sub_sat(xu_mul_sat(i, h), xu_mul_sat(r, l));
  r = t;
}
FX_HD void xf_rot_c(int32_t &r, int32_t &i, int32_t h, int32_t l) { /* :1853-1856 */
  const int32_t t = fx_sub_sat(xu_mul_sat(i, h), xu_mul_sat(r, l));
  i = fx_add_sat(xu_mul_sat(r, h), xu_mul_sat(i, l));
  r = t;
}
/* the butterfly of every pass (:1476-1501); alt: the last twiddle quadrant's (:1866-1883) */
FX_HD void xf_fwd_bfly(int32_t *p0, int32_t *p1, int32_t *p2, int32_t *p3, bool alt, int32_t *o0, int32_t *o1, int32_t *o2, int32_t *o3) {
  int32_t x0r = p0[0], x0i = p0[1], x1r = p1[0], x1i = p1[1], x2r = p2[0], x2i = p2[1], x3r = p3[0], x3i = p3[1];
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
  x2r = fx_add_sat(x2r, x3i);
  x2i = fx_sub_sat(x2i, x3r);
  x3i = fx_sub_sat(x2r, xu_shl1(x3i));
  x3r = fx_add_sat(x2i, xu_shl1(x3r));
  o0[0] = x0r; o0[1] = x0i;
  o1[0] = x2r; o1[1] = x2i;
  o2[0] = x1r; o2[1] = x1i;
  o3[0] = x3i; o3[1] = x3r;
}
FX_HD int xf_fft_fwd_p2(int32_t *xr, int32_t *xi, int npoints, int32_t *y) {
  int n = 0;
  while ((npoints >> (n + 1)) != 0) n++;
  const int not_power_4 = n & 1;
  int n_stages = n >> 1;
  int shift = (n % 2 == 0) ? (n + 4) / 2 : (n + 3) / 2;
  const int dig_rev_shift = 15 - n; /* norm32(npoints) + 1 - 16 */
  const int32_t *w = xaac_usac_fft_tw;
  for (int i = 0; i < npoints; i += 4) { /* first pass: inputs divided (C's truncating division, :1443), digit-reversed */
    unsigned h2 = xu_dig_rev((unsigned)i, dig_rev_shift);
    if (not_power_4) h2 = (h2 + 1) & ~1u;
    int32_t p[4][2];
    for (int q = 0; q < 4; q++) {
      const int c = (int)(h2 >> 1) + q * (npoints >> 2); /* legs npoints / 2 words = npoints / 4 complex points apart */
      p[q][0] = xu_div_pow2(xr[c], shift);
      p[q][1] = xu_div_pow2(xi[c], shift);
    }
    xf_fwd_bfly(p[0], p[1], p[2], p[3], false, y + 2 * i, y + 2 * i + 2, y + 2 * i + 4, y + 2 * i + 6);
  }
  int del = 4, nodespacing = 64, in_loop_cnt = npoints >> 4;
  for (int st = n_stages - 1; st > 0; st--) {
    const int S = nodespacing * del;
    const int sec = S / 4 + S / 8 - S / 16 + S / 32 - S / 64 + S / 128 - S / 256; /* :1566-1570 */
    for (int jj = 0; jj < del; jj++) {
      const int j = jj * nodespacing;
      /* which of the loops at :1505 (no twiddles), :1573, :1656, :1733, :1815 the column falls into */
      const int quad = jj == 0 ? 0 : (j <= sec ? 1 : (j <= (S >> 1) ? 2 : (j <= 2 * sec ? 3 : 4)));
      for (int k = 0; k < in_loop_cnt; k++) {
        int32_t *d0 = y + 2 * (4 * del * k + jj), *d1 = d0 + 2 * del, *d2 = d1 + 2 * del, *d3 = d2 + 2 * del;
        int32_t a[2] = {d0[0], d0[1]}, b[2] = {d1[0], d1[1]}, c[2] = {d2[0], d2[1]}, d[2] = {d3[0], d3[1]};
        if (quad) {
          xf_rot_a(b[0], b[1], w[2 * j], w[2 * j + 1]);
          if (quad <= 2) xf_rot_a(c[0], c[1], w[4 * j], w[4 * j + 1]);
          else xf_rot_b(c[0], c[1], w[4 * j - 512], w[4 * j - 511]);
          if (quad == 1) xf_rot_a(d[0], d[1], w[6 * j], w[6 * j + 1]);
          else if (quad <= 3) xf_rot_b(d[0], d[1], w[6 * j - 512], w[6 * j - 511]);
          else xf_rot_c(d[0], d[1], w[6 * j - 1024], w[6 * j - 1023]);
        }
        xf_fwd_bfly(a, b, c, d, quad == 4, d0, d1, d2, d3);
      }
    }
    nodespacing >>= 2;
    del <<= 2;
    in_loop_cnt >>= 2;
  }
  if (not_power_4) { /* the radix-2 stage, :1903-1963 */
    nodespacing <<= 1;
    shift += 1;
    for (int q = 0; q < del; q++) { /* complex points q and q + del */
      const int jt = (q < del / 2 ? q : q - del / 2) * nodespacing * 2;
      const int32_t w1h = w[jt], w1l = w[jt + 1];
      const int32_t x0r = y[2 * q], x0i = y[2 * q + 1];
      int32_t x1r = y[2 * (q + del)], x1i = y[2 * (q + del) + 1];
      if (q < del / 2) xf_rot_a(x1r, x1i, w1h, w1l);
      else xf_rot_b(x1r, x1i, w1h, w1l);
      y[2 * (q + del)] = x0r / 2 - x1r / 2;
      y[2 * (q + del) + 1] = x0i / 2 - x1i / 2;
      y[2 * q] = x0r / 2 + x1r / 2;
      y[2 * q + 1] = x0i / 2 + x1i / 2;
    }
  }
  for (int i = 0; i < npoints; i++) {
    xr[i] = y[2 * i];
    xi[i] = y[2 * i + 1];
  }
  return shift;
}
/* ixheaacd_complex_fft_p3, fft_mode = -1 (fft.c:2531): nlength = 3 * 2^k; tr / ti: nlength / 3 words of scratch each */
FX_HD int xf_fft_fwd_p3(int32_t *xr, int32_t *xi, int nlength, int32_t *y, int32_t *tr, int32_t *ti) {
  const int mpass = nlength / 3;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < mpass; j++) {
      tr[j] = xr[3 * j + i];
      ti[j] = xi[3 * j + i];
    }
    xf_fft_fwd_p2(tr, ti, mpass, y);
    for (int j = 0; j < mpass; j++) {
      xr[3 * j + i] = tr[j];
      xi[3 * j + i] = ti[j];
    }
  }
  int n = 0;
  while ((mpass >> (n + 1)) != 0) n++;
  const int shift = (n % 2 == 0) ? (n + 4) / 2 : (n + 5) / 2;
  int idx = 0;
  for (int g = 0; g < mpass; g++) { /* a group: points 3 g .. 3 g + 2 halved, two rotated (:2586-2607), the 3-point butterfly (:2493) */
    int32_t in[6];
    for (int q = 0; q < 3; q++) {
      in[2 * q] = xr[3 * g + q] >> 1;
      in[2 * q + 1] = xi[3 * g + q] >> 1;
    }
    for (int q = 1; q < 3; q++) {
      idx++;
      const int32_t c = xaac_usac_tw3_r[idx], sn = xaac_usac_tw3_i[idx];
      const int32_t t = fx_sub_sat(xu_mul_sat(in[2 * q], c), xu_mul_sat(in[2 * q + 1], sn));
      in[2 * q + 1] = fx_add_sat(xu_mul_sat(in[2 * q], sn), xu_mul_sat(in[2 * q + 1], c));
      in[2 * q] = t;
    }
    idx += 3 * (128 / mpass - 1) + 1;
    const int32_t sinmu = 1859775393; /* -1859775393 * sign_dir */
    const int32_t temp_real = fx_add_sat(in[0], in[2]), temp_imag = fx_add_sat(in[1], in[3]);
    const int32_t add_r = fx_add_sat(in[2], in[4]), add_i = fx_add_sat(in[3], in[5]);
    const int32_t sub_r = fx_sub_sat(in[2], in[4]), sub_i = fx_sub_sat(in[3], in[5]);
    const int32_t p1 = add_r >> 1, p4 = add_i >> 1, p2 = xu_mul32_shl(sub_i, sinmu), p3 = xu_mul32_shl(sub_r, sinmu);
    const int32_t temp = fx_sub(in[0], p1);
    y[6 * g] = fx_add_sat(temp_real, in[4]);
    y[6 * g + 1] = fx_add_sat(temp_imag, in[5]);
    y[6 * g + 2] = fx_add_sat(temp, p2);
    y[6 * g + 3] = fx_sub_sat(fx_sub_sat(in[1], p3), p4);
    y[6 * g + 4] = fx_sub_sat(temp, p2);
    y[6 * g + 5] = fx_sub_sat(fx_add_sat(in[1], p3), p4);
  }
  for (int i = 0; i < mpass; i++) {
    xr[i] = y[6 * i];
    xi[i] = y[6 * i + 1];
    xr[mpass + i] = y[6 * i + 2];
    xi[mpass + i] = y[6 * i + 3];
    xr[2 * mpass + i] = y[6 * i + 4];
    xi[2 * mpass + i] = y[6 * i + 5];
  }
  return shift + 1; /* *preshift = shift - *preshift + 1 with *preshift = 0 (the inner transforms report into a local) */
}

/* ---- ixheaacd_cal_fac_data (imdct.c:210).  in: what the reference reads from usac_data for the channel -- fac_data[0 .. lfac],
   lpc_prev[0 .. 16], acelp_in[0 .. n_long / 4); n_long = ccfl; lfac as ixheaacd_fd_frm_dec chooses it (:620-632).
   fac_idata: 2 lfac words out (the caller's array behind its first 16); *q_fac: the exponent.  Returns 0, or -1 where the
   reference does (ec_flag 0).  All lanes of the team call it; w is the team's. --------------------------------------------- */
template <class In>
FX_HD int xf_cal_fac_data(const XfCx &cx, XfWork *w, const In *in, int n_long, int lfac, int32_t *fac_idata, int32_t *q_fac) {
  const int nz = n_long / 4;
  XF_ONE {
    const int32_t g0 = in->fac_data[0];
    const int quo = g0 / 28, rem = g0 % 28;
    float pow10 = 1;
    for (int q = quo; q > 0; q--) pow10 *= 10;
    const float rem10 = xf_bits(xaac_usac_pow10_f32_bits[rem < 0 ? 0 : (rem > 27 ? 27 : rem)]);
    const float gain = pow10 * rem10;
    int scale = fx_norm32(fx_f2i_trunc((gain < 0 ? -gain : gain) + 1));
    w->s_gain_fac = fx_f2i_trunc(gain * xf_pow2(scale));
    w->s_scale = scale + 4;
    w->s_qfac1 = 1.0f / gain;
    w->s_err = 0;
  }
  XF_PAR(k, 0, 264) w->izir[k] = 0;
  XF_PAR(k, 0, 16 + 256) w->out[k] = 0;
  cx.sync();
  const float qfac1 = w->s_qfac1;
  { /* :238-253: the zero-input response as integers at the largest exponent its peak allows */
    float m = 0.0f;
    XF_PAR(k, 0, nz) {
      const float z = in->acelp_in[k] * qfac1, az = z < 0 ? -z : z;
      if (az > m) m = az;
    }
    if (cx.lane < 64) w->red[cx.lane] = m;
    cx.sync();
    XF_ONE {
      float ft = 0.0f; /* (the reference's running maximum: `if (ABS(z) > ftemp)` in index order; a maximum of maxima is the same number) */
      for (int l = 0; l < (cx.n < 64 ? cx.n : 64); l++)
        if (w->red[l] > ft) ft = w->red[l];
      w->s_itemp = fx_f2i_trunc(ft);
      w->s_q3 = fx_norm32(w->s_itemp);
    }
    cx.sync();
    const float sc = xf_pow2(w->s_q3);
    XF_PAR(k, 0, nz) w->izir[k] = fx_f2i_trunc((in->acelp_in[k] * qfac1) * sc);
  }
  cx.sync();
  XF_ONE { /* :255-270: the previous LPC filter likewise; :272-281: the FAC coefficients, whose peak search starts from the filter's */
    float ft = 0.0f;
    for (int k = 0; k <= 16; k++) {
      const float a = in->lpc_prev[k] < 0 ? -in->lpc_prev[k] : in->lpc_prev[k];
      if (a > ft) ft = a;
    }
    int32_t itemp = fx_f2i_trunc(ft);
    w->s_q2 = fx_norm32(itemp);
    const float sc = xf_pow2(w->s_q2);
    for (int k = 0; k <= 16; k++) w->aq[k] = fx_f2i_trunc(in->lpc_prev[k] * sc);
    for (int k = 0; k < lfac && k < 128; k++) {
      const int32_t a = fx_abs_sat(in->fac_data[k + 1]);
      if (a > itemp) itemp = a;
    }
    w->s_q1 = fx_norm32(itemp);
  }
  cx.sync();
  {
    const float sc = xf_pow2(w->s_q1);
    XF_PAR(k, 0, (lfac < 128 ? lfac : 128)) w->scaled[k + 1] = fx_f2i_trunc((float)in->fac_data[k + 1] * sc);
  }
  cx.sync();
  XF_PAR(k, 0, (lfac < 256 ? lfac : 256) / 2) { /* :283-286 */
    w->x_in[k] = w->scaled[2 * k + 1];
    w->x_in[lfac / 2 + k] = w->scaled[lfac - 2 * k];
  }
  cx.sync();
  /* :288-319 (ec_flag 0) */
  if (lfac > 128 || (n_long / 8) < lfac || (n_long / 8 + 1) > (2 * 256 - lfac - 1)) return -1;
  if ((lfac & (lfac - 1)) && lfac != 48 && lfac != 96 && lfac != 192 && lfac != 384 && lfac != 768) return -1;
  if (lfac != 48 && lfac != 64 && lfac != 96 && lfac != 128) return -1; /* lengths ixheaacd_acelp_mdct has no table of its own for (it would
                                                                            take the 24-point one): no frame ixheaacd_fd_frm_dec makes */
  const int nl = lfac / 2;
  const int32_t *tw = lfac == 48 ? xaac_usac_fac_tw_24 : (lfac == 64 ? xaac_usac_fac_tw_32 : (lfac == 96 ? xaac_usac_fac_tw_48 : xaac_usac_fac_tw_64));
  const int32_t *win = lfac == 48 ? xaac_usac_sine_win_96 : (lfac == 64 ? xaac_usac_sine_win_128 : (lfac == 96 ? xaac_usac_sine_win_192 : xaac_usac_sine_win_256));
  int32_t *fo = w->out + 16;
  XF_PAR(i, 0, nl) { /* ixheaacd_pre_twid */
    w->re[i] = xf_mul_sub64(w->x_in[i], w->x_in[nl + i], tw[i], tw[nl + i]);
    w->im[i] = xf_mul_add64(w->x_in[i], w->x_in[nl + i], tw[nl + i], tw[i]);
  }
  cx.sync();
  XF_ONE {
    int pre = (nl & (nl - 1)) ? xf_fft_fwd_p3(w->re, w->im, nl, w->y, w->x_in, w->x_in + 32) : xf_fft_fwd_p2(w->re, w->im, nl, w->y);
    w->s_preshift = pre + 2; /* acelp_mdct.c:212, :216 */
  }
  cx.sync();
  XF_PAR(i, 0, nl) { /* ixheaacd_post_twid */
    fo[2 * i] = xf_mul_sub64(w->re[i], w->im[i], tw[2 * nl + i], tw[3 * nl + i]);
    fo[2 * nl - 1 - 2 * i] = (int32_t)(0u - (uint32_t)xf_mul_add64(w->re[i], w->im[i], tw[3 * nl + i], tw[2 * nl + i]));
  }
  cx.sync();
  XF_ONE { /* the weighted filter and the recursion through it (fwd_alias_cnx.c:60-94); fo[lfac ..] is zero, fo[-16 .. -1] too */
    w->ap[0] = w->aq[0];
    int32_t f = 1975684956; /* IGAMMA1 */
    for (int i = 1; i <= 16; i++) {
      w->ap[i] = xf_mult32_m(f, w->aq[i]);
      f = xf_mult32_m(f, 1975684956);
    }
    const int q2 = w->s_q2;
    for (int i = 0; i < 2 * lfac; i++) {
      int32_t s = fo[i];
      for (int j = 1; j <= 16; j++) s = fx_sub_sat(s, xf_mul32_sh(w->ap[j], fo[i - j], q2));
      fo[i] = s;
    }
    w->s_preshift += 1;
  }
  cx.sync();
  { /* the zero-input response through the window's two slopes (fwd_alias_cnx.c:175-200) */
    const int sh = (int)(int8_t)(w->s_q3 - w->s_q1 + 31 + (int8_t)w->s_preshift);
    const int half = nz / 2;
    XF_PAR(i, 0, lfac) {
      const int32_t w_hi = 2147483647 - xf_mult32_m(win[lfac + i], win[lfac + i]);
      const int32_t w_lo = xf_mult32_m(win[lfac - 1 - i], win[2 * lfac - 1 - (lfac - 1 - i)]);
      const int32_t t1 = xf_mul32_sh(w->izir[1 + half + i], w_hi, sh);
      const int32_t t2 = xf_mul32_sh(w->izir[1 + half - 1 - i], w_lo, sh);
      const int64_t sum = (int64_t)(fo[i] / 2) + t1 + t2;
      fo[i] = xf_sat64(sum);
      fo[lfac + i] = fo[lfac + i] / 2;
    }
  }
  cx.sync();
  const int preshift = w->s_preshift + 4;
  XF_ONE *q_fac = w->s_q_out = (int32_t)(int8_t)(w->s_q1 - preshift);
  XF_PAR(k, 0, 2 * lfac) fac_idata[k] = xf_mul32_sh(fo[k], w->s_gain_fac, (int)(int8_t)w->s_scale);
  cx.sync();
  return 0;
}

#endif
