/*
 * hbe_trans.h -- the QMF-domain harmonic transposer between its two banks (hbe_poly.h) as host/device code: the
 * stretch-by-2 / 3 / 4 products of ixheaacd_hbe_post_anal_process (decoder/ixheaacd_hbe_trans.c:1549-1606 with
 * :298-372, :753-794, :796-1027, :1029-1082; the variants with a pitch below) and the frame bookkeeping of ixheaacd_qmf_hbe_apply
 * (:224-296).  Included by the oracle (oracle/oracle_hbe.cpp) and by hbe_kernel.hip; no contraction, the reference's
 * operand order and width (FLOAT32, with the FLOAT64 bases and libm calls where the reference has them).
 *
 * The reference walks output bands, and inside columns i = 0..15, each column adding a block of 10 / 8 / 6 products to
 * consecutive rows of qmf_out_buf, two rows further per column.  A block is a pure function of qmf_in_buf
 * (xh_prod2_block, xh_prod3_block, xh_prod4_block); a row's element then adds the blocks that reach it in the order of
 * the columns (xh_prod_gather).  Bands of the three stretch factors are disjoint.
 *
 * With a pitch (pitch_in_bins / 12 >= 1, :1572-1603) every column adds, after its block, one cross product of two
 * sub-bands a pitch apart to the two rows 2 i + 5 and 2 i + 6 (ixheaacd_hbe_post_anal_xprod2, :1084-1247, and
 * ixheaacd_hbe_xprod_proc_3 / _4, :374-567 / :569-751): xh_xprod2/3/4 compute the two terms, the gather adds them
 * behind the column's block.
 *
 * Not built: the 4:1 system, the DFT transposer.
 */
#ifndef XAAC_HBE_TRANS_H
#define XAAC_HBE_TRANS_H

#include <math.h>

#include "hbe_poly.h"

#pragma clang fp contract(off)

#define XH_ZERO_BAND 6 /* HBE_ZERO_BAND_IDX */

/* cbrt(double) as this platform's C library computes it (glibc sysdeps/ieee754/dbl-64/s_cbrt.c: a degree-6 polynomial
   of the mantissa, one Halley step, the exponent's third by table) -- restated because the GPU's own cbrt rounds
   differently; tests/test_hbe_oracle_vs_reference.py pins it on the host's cbrt. */
FX_HD double xh_cbrt(double x) {
  const double factor[5] = {1.0 / 1.5874010519681994748, 1.0 / 1.2599210498948731648, 1.0, 1.2599210498948731648,
                            1.5874010519681994748};
  int xe;
  const double xm = frexp(fabs(x), &xe);
  if (xe == 0 && (x == 0.0 || x != x || fabs(x) > 1.7976931348623157e308)) return x + x;
  const double u =
      (0.354895765043919860 +
       ((1.50819193781584896 -
         ((2.11499494167371287 -
           ((2.44693122563534430 - ((1.83469277483613086 - (0.784932344976639262 - 0.145263899385486377 * xm) * xm) * xm)) *
            xm)) *
          xm)) *
        xm));
  const double t2 = u * u * u;
  const double ym = u * (t2 + 2.0 * xm) / (2.0 * t2 + xm) * factor[2 + xe % 3];
  return ldexp(x > 0.0 ? ym : -ym, xe / 3);
}
/* (FLOAT32)ixheaac_cbrt_calc((FLOAT32)base): cbrt(1.0f / a), common/ixheaac_basic_op.h:25 */
FX_HD float xh_cbrt_inv(double base) {
  const float a = (float)base;
  return (float)xh_cbrt((double)(1.0f / a));
}

struct XhC {
  float r, i;
};
/* ixheaacd_norm_qmf_in_buf_2 (:337-372) on one sample */
FX_HD XhC xh_norm2(float x_r, float x_i) {
  double base = 1e-17;
  float temp = x_r * x_r;
  base = base + temp;
  temp = x_i * x_i;
  base = base + temp;
  float mag = (float)(1.0f / base);
  mag = (float)sqrt(sqrt((double)mag));
  const XhC y = {x_r * mag, x_i * mag};
  return y;
}
/* ixheaacd_norm_qmf_in_buf_4 (:298-335) on one sample */
FX_HD XhC xh_norm4(float x_r, float x_i) {
  double base = 1e-17;
  float temp = x_r * x_r;
  base = base + temp;
  temp = x_i * x_i;
  base = base + temp;
  temp = (float)sqrt(sqrt(base));
  float mag = temp * (float)(sqrt((double)temp));
  mag = 1 / mag;
  const XhC y = {x_r * mag, x_i * mag};
  return y;
}
/* the third-root normalisation of the stretch-by-3 products (:829-835) */
FX_HD XhC xh_norm3(float x_r, float x_i) {
  const double base = 1e-17;
  double base1 = base + x_r * x_r;
  base1 = base1 + x_i * x_i;
  const float mag = xh_cbrt_inv(base1);
  const XhC y = {x_r * mag, x_i * mag};
  return y;
}

/* in(row, band): (re, im) of qmf_in_buf[row][2 band ..].
   The products work on normalised samples; a normalised sample is a function of (row, input band) alone, so the
   *_n forms take accessors for them (the kernel computes each once per frame and keeps it in LDS, the plain forms
   normalise on the fly). */
/* Stretch by 2 (:753-794): the ten products column i adds to rows 1 + 2 i .. of band qb; blk[2 k ..] = product k.
   n2(row) = xh_norm2 of the band's sample in `row`. */
template <class N2>
FX_HD void xh_prod2_block_n(const N2 &n2, int i, float *blk) {
  const XhC z = n2(XH_ZERO_BAND + i);
  for (int k = 0; k < 10; k++) {
    const XhC n = n2(1 + i + k);
    blk[2 * k] = ((n.r * z.r - n.i * z.i) * 0.3333333f);
    blk[2 * k + 1] = ((n.r * z.i + n.i * z.r) * 0.3333333f);
  }
}
template <class In>
FX_HD void xh_prod2_block(const In &in, int qb, int i, float *blk) {
  xh_prod2_block_n([&](int row) { const XhC v = in(row, qb); return xh_norm2(v.r, v.i); }, i, blk);
}
/* Stretch by 4 (:1029-1082): six products, rows 3 + 2 i ..; nz(row) / nn(row) = xh_norm4 of band qb >> 1 / of its
   neighbour (above for odd qb, below for even) */
template <class NZ, class NN>
FX_HD void xh_prod4_block_n(const NZ &nz, const NN &nn, int i, float *blk) {
  const XhC z = nz(XH_ZERO_BAND + i);
  const float temp_r = z.r, temp_i = z.i;
  float zr = z.r, zi = z.i;
  const float temp = zr * zr - zi * zi;
  zi = zr * zi + zi * zr;
  zr = temp_r * temp - temp_i * zi;
  zi = temp_r * zi + temp_i * temp;
  for (int k = 0; k < 6; k++) {
    const XhC n = nn(i + 2 * k);
    blk[2 * k] = ((n.r * zr - n.i * zi) * 0.6666667f);
    blk[2 * k + 1] = ((n.r * zi + n.i * zr) * 0.6666667f);
  }
}
template <class In>
FX_HD void xh_prod4_block(const In &in, int qb, int i, float *blk) {
  const int inp = qb >> 1, ip = (qb & 1) ? inp + 1 : inp - 1;
  xh_prod4_block_n([&](int row) { const XhC v = in(row, inp); return xh_norm4(v.r, v.i); },
                   [&](int row) { const XhC v = in(row, ip); return xh_norm4(v.r, v.i); }, i, blk);
}
/* Stretch by 3 (:796-1027): eight products, rows 2 + 2 i ..; the eight inputs alternate between a sample (rows i + 3 m)
   and a point interpolated from the two rows behind it, both normalised by the cube root.
   xh_interp3: the interpolated point of input band b behind row r, before normalisation; the reference adds its four
   terms in one order where the output band reads one input band (2 qb mod 3 < 2, :842-851) and in another where it
   reads two (:902-925). */
template <class In>
FX_HD XhC xh_interp3(const In &in, int band, int r, bool two_band_order) {
  const float t[5][8] = {{1, -1, 1, 1, 1, 1, -1, 1}, {1, 1, -1, 1, 1, -1, 1, 1}, {-1, 1, -1, -1, -1, -1, 1, -1},
                         {-1, -1, 1, -1, -1, 1, -1, -1}, {1, -1, 1, 1, 1, 1, -1, 1}}; /* ixheaac_sel_case (esbr_rom.c:3128) */
  const float *sel = t[(band + 1) & 3]; /* the second band of a pair takes the next row, which is this formula again */
  const XhC b = in(r + 2, band), c = in(r + 1, band);
  float tr = sel[0] * b.r + sel[1] * b.i, ti = sel[2] * b.r + sel[3] * b.i;
  if (two_band_order) {
    tr = tr + sel[4] * c.r + sel[5] * c.i;
    ti = ti + sel[6] * c.r + sel[7] * c.i;
  } else {
    tr += sel[4] * c.r + sel[5] * c.i;
    ti += sel[6] * c.r + sel[7] * c.i;
  }
  tr *= 0.3984033437f;
  ti *= 0.3984033437f;
  const XhC y = {tr, ti};
  return y;
}
/* na(band_sel, row) = xh_norm3 of the sample, nb(band_sel, row) = xh_norm3 of xh_interp3 behind row; band_sel 0: input
   band 2 qb / 3, 1: the band above it (read when 2 qb mod 3 == 2) */
template <class NA, class NB>
FX_HD void xh_prod3_block_n(const NA &na, const NB &nb, int rem, int i, float *blk) {
  if (rem == 0 || rem == 1) {
    XhC vec[8];
    for (int m = 0; m < 4; m++) {
      vec[2 * m] = na(0, i + 3 * m);
      vec[2 * m + 1] = nb(0, i + 3 * m);
    }
    const float tr = vec[XH_ZERO_BAND - 2].r, ti = vec[XH_ZERO_BAND - 2].i;
    const float zr = tr * tr - ti * ti, zi = tr * ti + ti * tr;
    for (int k = 0; k < 8; k++) {
      const float pr = vec[k].r * zr - vec[k].i * zi, pi = vec[k].r * zi + vec[k].i * zr;
      blk[2 * k] = (pr * 0.4714045f);
      blk[2 * k + 1] = (pi * 0.4714045f);
    }
  } else {
    XhC vec[8], cap[8];
    for (int m = 0; m < 4; m++) {
      vec[2 * m] = na(1, i + 3 * m);
      cap[2 * m] = na(0, i + 3 * m);
      vec[2 * m + 1] = nb(1, i + 3 * m);
      cap[2 * m + 1] = nb(0, i + 3 * m);
    }
    float tr = cap[XH_ZERO_BAND - 2].r, ti = cap[XH_ZERO_BAND - 2].i;
    const float tr1 = vec[XH_ZERO_BAND - 2].r, ti1 = vec[XH_ZERO_BAND - 2].i;
    const float zr = tr * tr - ti * ti, zi = tr * ti + ti * tr;
    tr = tr1 * tr1 - ti1 * ti1;
    ti = tr1 * ti1 + ti1 * tr1;
    for (int k = 0; k < 8; k++) {
      float pr = vec[k].r * zr - vec[k].i * zi, pi = vec[k].r * zi + vec[k].i * zr;
      pr += cap[k].r * tr - cap[k].i * ti;
      pi += cap[k].r * ti + cap[k].i * tr;
      blk[2 * k] = (pr * 0.23570225f);
      blk[2 * k + 1] = (pi * 0.23570225f);
    }
  }
}
template <class In>
FX_HD void xh_prod3_block(const In &in, int qb, int i, float *blk) {
  const int inp = (2 * qb) / 3, rem = 2 * qb - 3 * inp;
  xh_prod3_block_n([&](int sel, int row) { const XhC v = in(row, inp + sel); return xh_norm3(v.r, v.i); },
                   [&](int sel, int row) { const XhC v = xh_interp3(in, inp + sel, row, rem == 2); return xh_norm3(v.r, v.i); },
                   rem, i, blk);
}

/* ---- the pitch-adaptive cross products ---------------------------------------------------------------------------- */
/* inf(row, idx): word idx of qmf_in_buf row `row` with the reference's flat addressing -- the candidate sub-bands of a
   cross product may lie before a row's first or behind its last pair, and the reference reads the neighbouring row
   there before it rejects the candidate (rows are contiguous in its memory as they are in xaac_hbe_state). */
FX_HD void xh_cpow(float &r, float &i, int n) { /* n - 1 further multiplications by the start value (:509-514) */
  const float temp_r = r, temp_i = i;
  for (int idx = 0; idx < n - 1; idx++) {
    const float tmp = r;
    r = r * temp_r - i * temp_i;
    i = tmp * temp_i + i * temp_r;
  }
}
/* Stretch by 2 (:1124-1240).  cross[0..3] = the terms of rows 2 i + 5 and 2 i + 6; returns whether there is one. */
template <class Inf>
FX_HD bool xh_xprod2(const Inf &inf, int qb, int i, float p, int pitch_idx, float *cross) {
  const double temp_fac = (2.0 * qb + 1 - p) * 0.5;
  const int n1 = ((int)(temp_fac)) << 1, n2 = ((int)(temp_fac + p)) << 1;
  const int row = i + XH_ZERO_BAND;
  const float mag_zero_band = inf(row, 2 * qb) * inf(row, 2 * qb) + inf(row, 2 * qb + 1) * inf(row, 2 * qb + 1);
  const float mag_n1_band = inf(row, n1) * inf(row, n1) + inf(row, n1 + 1) * inf(row, n1 + 1);
  const float mag_n2_band = inf(row, n2) * inf(row, n2) + inf(row, n2 + 1) * inf(row, n2 + 1);
  const float temp = mag_n1_band < mag_n2_band ? mag_n1_band : mag_n2_band;
  float max_mag_value = 0;
  int max_n1 = 0, max_n2 = 0;
  if (temp > 0) {
    max_mag_value = temp;
    max_n1 = n1;
    max_n2 = n2;
  }
  if (!(max_mag_value > mag_zero_band && max_n1 >= 0 && max_n2 < 128)) return false;
  const XhC z = xh_norm2(inf(row, max_n1), inf(row, max_n1 + 1)); /* mid_trans_fac = 1: no further power */
  XhC y[2];
  for (int k = 0; k < 2; k++) y[k] = xh_norm2(inf(row - 1 + k, max_n2), inf(row - 1 + k, max_n2 + 1));
  const float cs0 = xaac_hbe_xprod_cs_2[(pitch_idx << 1) + 0], cs1 = xaac_hbe_xprod_cs_2[(pitch_idx << 1) + 1];
  const float mag_cmplx_gain = 1.666666667f;
  float temp_r = y[0].r * z.r - y[0].i * z.i, temp_i = y[0].r * z.i + y[0].i * z.r;
  const float tmp_r1 = (float)(cs0 * temp_r - cs1 * temp_i);
  temp_i = (float)(cs0 * temp_i + cs1 * temp_r);
  cross[0] = (float)(mag_cmplx_gain * tmp_r1);
  cross[1] = (float)(mag_cmplx_gain * temp_i);
  temp_r = y[1].r * z.r - y[1].i * z.i;
  temp_i = y[1].r * z.i + y[1].i * z.r;
  cross[2] = (float)(mag_cmplx_gain * temp_r);
  cross[3] = (float)(mag_cmplx_gain * temp_i);
  return true;
}
/* the common tail of ixheaacd_hbe_xprod_proc_3 / _4: powers, product, rotation of the first term, gain */
FX_HD void xh_xprod_tail(float zr, float zi, float *yr, float *yi, int mid_trans_fac, int max_trans_fac, float cos_theta,
                         float sin_theta, float gain, float *cross) {
  xh_cpow(zr, zi, mid_trans_fac);
  for (int k = 0; k < 2; k++) xh_cpow(yr[k], yi[k], max_trans_fac);
  float o_r[2], o_i[2];
  for (int k = 0; k < 2; k++) {
    o_r[k] = yr[k] * zr - yi[k] * zi;
    o_i[k] = yr[k] * zi + yi[k] * zr;
  }
  const float temp_r = o_r[0], temp_i = o_i[0];
  o_r[0] = (float)(cos_theta * temp_r - sin_theta * temp_i);
  o_i[0] = (float)(cos_theta * temp_i + sin_theta * temp_r);
  for (int k = 0; k < 2; k++) {
    cross[2 * k] = (float)(gain * o_r[k]);
    cross[2 * k + 1] = (float)(gain * o_i[k]);
  }
}
/* Stretch by 3 (ixheaacd_hbe_xprod_proc_3, :374-567) */
template <class Inf>
FX_HD bool xh_xprod3(const Inf &inf, int qb, int i, float p, int pitch_idx, float *cross) {
  const int inp = 2 * qb / 3, row = i + XH_ZERO_BAND;
  const float mag_zero_band = inf(row, 2 * inp) * inf(row, 2 * inp) + inf(row, 2 * inp + 1) * inf(row, 2 * inp + 1);
  float max_mag_value = 0;
  int max_n1 = 0, max_n2 = 0, max_trans_fac = 0;
  for (int tr = 1; tr < 3; tr++) {
    const double temp_fac = (2.0f * qb + 1 - tr * p) * 0.3333334;
    const int n1 = (int)(temp_fac), n2 = (int)(temp_fac + p);
    const float mag_n1_band = inf(row, 2 * n1) * inf(row, 2 * n1) + inf(row, 2 * n1 + 1) * inf(row, 2 * n1 + 1);
    const float mag_n2_band = inf(row, 2 * n2) * inf(row, 2 * n2) + inf(row, 2 * n2 + 1) * inf(row, 2 * n2 + 1);
    const float temp = mag_n1_band < mag_n2_band ? mag_n1_band : mag_n2_band;
    if (temp > max_mag_value) {
      max_mag_value = temp;
      max_trans_fac = tr;
      max_n1 = n1;
      max_n2 = n2;
    }
  }
  if (!(max_mag_value > mag_zero_band && max_n1 >= 0 && max_n2 < 64)) return false;
  int mid_trans_fac = 3 - max_trans_fac;
  float d1, d2;
  int nz, ny; /* the sub-band of the zero-band factor, the sub-band of the two-row vector */
  if (max_trans_fac == 1) {
    d1 = 0;
    d2 = 1.5;
    nz = max_n1;
    ny = max_n2;
  } else {
    d1 = 1.5;
    d2 = 0;
    mid_trans_fac = max_trans_fac;
    max_trans_fac = 3 - max_trans_fac;
    nz = max_n2;
    ny = max_n1;
  }
  float zr = inf(row, 2 * nz), zi = inf(row, 2 * nz + 1);
  const int idx = ((ny & 3) + 1) & 3;
  const float c0r = xaac_hbe_interp_coeff[2 * idx], c0i = xaac_hbe_interp_coeff[2 * idx + 1];
  const float c1r = c0r, c1i = -c0i;
  float yr[2], yi[2];
  yr[1] = inf(row, 2 * ny);
  yi[1] = inf(row, 2 * ny + 1);
  float temp_r = inf(row - 2, 2 * ny), temp_i = inf(row - 2, 2 * ny + 1);
  yr[0] = c1r * temp_r - c1i * temp_i;
  yi[0] = c1i * temp_r + c1r * temp_i;
  temp_r = inf(row - 1, 2 * ny);
  temp_i = inf(row - 1, 2 * ny + 1);
  yr[0] += c0r * temp_r - c0i * temp_i;
  yi[0] += c0i * temp_r + c0r * temp_i;
  {
    const XhC z = xh_norm3(zr, zi);
    zr = z.r;
    zi = z.i;
    for (int k = 0; k < 2; k++) {
      const XhC y = xh_norm3(yr[k], yi[k]);
      yr[k] = y.r;
      yi[k] = y.i;
    }
  }
  const float cos_theta = xaac_hbe_xprod_cs_3[(pitch_idx << 1) + 0];
  float sin_theta = xaac_hbe_xprod_cs_3[(pitch_idx << 1) + 1];
  if (d2 < d1) sin_theta = -sin_theta;
  xh_xprod_tail(zr, zi, yr, yi, mid_trans_fac, max_trans_fac, cos_theta, sin_theta, 1.8856f, cross);
  return true;
}
/* Stretch by 4 (ixheaacd_hbe_xprod_proc_4, :569-751); n1 / n2 are word indices here as in the reference */
template <class Inf>
FX_HD bool xh_xprod4(const Inf &inf, int qb, int i, float p, int pitch_idx, float *cross) {
  const int inp = qb >> 1, row = i + XH_ZERO_BAND;
  const float mag_zero_band = inf(row, 2 * inp) * inf(row, 2 * inp) + inf(row, 2 * inp + 1) * inf(row, 2 * inp + 1);
  float max_mag_value = 0;
  int max_n1 = 0, max_n2 = 0, max_trans_fac = 0;
  for (int tr = 1; tr < 4; tr++) {
    const double temp_fac = (2.0 * qb + 1 - tr * p) * 0.25;
    const int n1 = ((int)(temp_fac)) << 1, n2 = ((int)(temp_fac + p)) << 1;
    const float mag_n1_band = inf(row, n1) * inf(row, n1) + inf(row, n1 + 1) * inf(row, n1 + 1);
    const float mag_n2_band = inf(row, n2) * inf(row, n2) + inf(row, n2 + 1) * inf(row, n2 + 1);
    const float temp = mag_n1_band < mag_n2_band ? mag_n1_band : mag_n2_band;
    if (temp > max_mag_value) {
      max_mag_value = temp;
      max_trans_fac = tr;
      max_n1 = n1;
      max_n2 = n2;
    }
  }
  if (!(max_mag_value > mag_zero_band && max_n1 >= 0 && max_n2 < 128)) return false;
  int mid_trans_fac = 4 - max_trans_fac;
  float d1, d2, zr, zi, yr[2], yi[2];
  if (max_trans_fac == 1) {
    d1 = 0;
    d2 = 2;
    zr = inf(row, max_n1);
    zi = inf(row, max_n1 + 1);
    for (int k = 0; k < 2; k++) {
      yr[k] = inf(row + 2 * (k - 1), max_n2);
      yi[k] = inf(row + 2 * (k - 1), max_n2 + 1);
    }
  } else if (max_trans_fac == 2) {
    d1 = 0;
    d2 = 1;
    zr = inf(row, max_n1);
    zi = inf(row, max_n1 + 1);
    for (int k = 0; k < 2; k++) {
      yr[k] = inf(row + (k - 1), max_n2);
      yi[k] = inf(row + (k - 1), max_n2 + 1);
    }
  } else {
    d1 = 2;
    d2 = 0;
    mid_trans_fac = max_trans_fac;
    max_trans_fac = 4 - max_trans_fac;
    zr = inf(row, max_n2);
    zi = inf(row, max_n2 + 1);
    for (int k = 0; k < 2; k++) {
      yr[k] = inf(row + 2 * (k - 1), max_n1);
      yi[k] = inf(row + 2 * (k - 1), max_n1 + 1);
    }
  }
  {
    const XhC z = xh_norm4(zr, zi);
    zr = z.r;
    zi = z.i;
    for (int k = 0; k < 2; k++) {
      const XhC y = xh_norm4(yr[k], yi[k]);
      yr[k] = y.r;
      yi[k] = y.i;
    }
  }
  float cos_theta, sin_theta;
  if (d2 == 1) {
    cos_theta = xaac_hbe_xprod_cs_4_1[(pitch_idx << 1) + 0];
    sin_theta = xaac_hbe_xprod_cs_4_1[(pitch_idx << 1) + 1];
  } else {
    cos_theta = xaac_hbe_xprod_cs_4[(pitch_idx << 1) + 0];
    sin_theta = xaac_hbe_xprod_cs_4[(pitch_idx << 1) + 1];
    if (d2 < d1) sin_theta = -sin_theta;
  }
  xh_xprod_tail(zr, zi, yr, yi, mid_trans_fac, max_trans_fac, cos_theta, sin_theta, 2.0f, cross);
  return true;
}
/* the pitch in sub-bands and whether the frame takes the cross products (:1558-1562) */
FX_HD float xh_pitch(int pitch_in_bins) { return (float)(pitch_in_bins * 0.08333333333333); }
#define XH_BLK 25 /* floats per (band, column): <= 10 complex products, two cross terms, whether they exist */
/* a column's whole contribution to band qb: blk[0..19] the block, blk[20..23] the cross terms, blk[24] != 0 with them */
/* the cross terms of a column: blk[20..23], blk[24] != 0 with them */
template <class Inf>
FX_HD void xh_column_cross(const Inf &inf, int factor, int qb, int i, int pitch_in_bins, float *blk) {
  const float p = xh_pitch(pitch_in_bins);
  bool has = false;
  if (!(p < 1.0f)) {
    if (factor == 2) has = xh_xprod2(inf, qb, i, p, pitch_in_bins, blk + 20);
    else if (factor == 3) has = xh_xprod3(inf, qb, i, p, pitch_in_bins, blk + 20);
    else has = xh_xprod4(inf, qb, i, p, pitch_in_bins, blk + 20);
  }
  blk[24] = has ? 1.0f : 0.0f;
}
template <class In, class Inf>
FX_HD void xh_column_block(const In &in, const Inf &inf, int factor, int qb, int i, int pitch_in_bins, float *blk) {
  if (factor == 2) xh_prod2_block(in, qb, i, blk);
  else if (factor == 3) xh_prod3_block(in, qb, i, blk);
  else xh_prod4_block(in, qb, i, blk);
  xh_column_cross(inf, factor, qb, i, pitch_in_bins, blk);
}
/* which stretch factor writes output band qb (0: none), from x_over_qmf and max_stretch (:1562-1580) */
FX_HD int xh_band_factor(const int32_t *xo, int max_stretch, int qb) {
  if (2 <= max_stretch && qb >= xo[0] && qb < xo[1]) return 2;
  if (3 <= max_stretch && qb >= xo[1] && qb < xo[2]) return 3;
  if (4 <= max_stretch && qb >= xo[2] && qb < xo[3]) return 4;
  return 0;
}
FX_HD int xh_block_len(int factor) { return factor == 2 ? 10 : (factor == 3 ? 8 : 6); }
FX_HD int xh_block_row0(int factor) { return factor - 1; } /* rows 1 / 2 / 3 + 2 i */

/* element `comp` (0 re, 1 im) of row r of an output band after the frame's products: `start` = what the row held
   before (the previous frame's rows 32.. moved down, zero for the upper half); blk(i)[..] = column i's block */
template <class Blk>
FX_HD float xh_prod_gather(float start, int factor, int r, int comp, const Blk &blk) {
  const int len = xh_block_len(factor), r0 = xh_block_row0(factor);
  float acc = start;
  /* only columns with 0 <= r - r0 - 2 i < len reach the row (the cross terms' column (r - 5) / 2 is one of them) */
  int i_lo = (r - r0 - len + 2) >> 1, i_hi = (r - r0) >> 1;
  if (i_lo < 0) i_lo = 0;
  if (i_hi > XAAC_HBE_NO_BINS / 2 - 1) i_hi = XAAC_HBE_NO_BINS / 2 - 1;
  for (int i = i_lo; i <= i_hi; i++) {
    const float *b = blk(i);
    const int k = r - r0 - 2 * i;
    if (k >= 0 && k < len) acc += b[2 * k + comp];
    const int kc = r - (2 * i + XH_ZERO_BAND - 1); /* the cross terms follow the column's block (:1118-1240) */
    if ((kc == 0 || kc == 1) && b[24] != 0.0f) acc += b[20 + 2 * kc + comp];
  }
  return acc;
}

/* both components of an element in one walk over the columns (each component's additions in the order of xh_prod_gather) */
template <class Blk>
FX_HD void xh_prod_gather2(float &re, float &im, int factor, int r, const Blk &blk) {
  const int len = xh_block_len(factor), r0 = xh_block_row0(factor);
  int i_lo = (r - r0 - len + 2) >> 1, i_hi = (r - r0) >> 1;
  if (i_lo < 0) i_lo = 0;
  if (i_hi > XAAC_HBE_NO_BINS / 2 - 1) i_hi = XAAC_HBE_NO_BINS / 2 - 1;
  for (int i = i_lo; i <= i_hi; i++) {
    const float *b = blk(i);
    const int k = r - r0 - 2 * i;
    if (k >= 0 && k < len) {
      re += b[2 * k];
      im += b[2 * k + 1];
    }
    const int kc = r - (2 * i + XH_ZERO_BAND - 1); /* the cross terms follow the column's block (:1118-1240) */
    if ((kc == 0 || kc == 1) && b[24] != 0.0f) {
      re += b[20 + 2 * kc];
      im += b[20 + 2 * kc + 1];
    }
  }
}

/* the frame's parameters are usable: bank sizes in the tables, cross-over bands inside the rows, a pitch inside its
   seven bits, the reference's own x_over_qmf[2] > 1 condition (:1573) */
FX_HD bool xh_apply_params_ok(const xaac_hbe_state *st, int pitch_in_bins) {
  const int s = st->synth_size, ks = st->k_start;
  if (!xh_size_ok(s) || ks < 0 || ks + s > 64 || ks * 32 + 2 * s > 7 * 64 || 4 * ks + 4 * s > 128) return false;
  if (st->start_band < 0 || st->end_band > 64 || st->max_stretch < 0 || st->max_stretch > 4) return false;
  if (pitch_in_bins < 0 || pitch_in_bins > 127) return false;
  for (int q = 0; q < st->max_stretch && q < 4; q++) {
    if (st->x_over_qmf[q] < 0 || st->x_over_qmf[q] > 64) return false;
    if (q && st->x_over_qmf[q] < st->x_over_qmf[q - 1]) return false;
  }
  if (st->max_stretch >= 4 && st->x_over_qmf[2] <= 1) return false;
  return true;
}

#endif /* XAAC_HBE_TRANS_H */
