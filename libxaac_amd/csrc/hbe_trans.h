/*
 * hbe_trans.h -- the QMF-domain harmonic transposer between its two banks (hbe_poly.h) as host/device code: the
 * stretch-by-2 / 3 / 4 products of ixheaacd_hbe_post_anal_process without a pitch (decoder/ixheaacd_hbe_trans.c:
 * 1549-1571 with :298-372, :753-794, :796-1027, :1029-1082) and the frame bookkeeping of ixheaacd_qmf_hbe_apply
 * (:224-296).  Included by the oracle (oracle/oracle_hbe.cpp) and by hbe_kernel.hip; no contraction, the reference's
 * operand order and width (FLOAT32, with the FLOAT64 bases and libm calls where the reference has them).
 *
 * The reference walks output bands, and inside columns i = 0..15, each column adding a block of 10 / 8 / 6 products to
 * consecutive rows of qmf_out_buf, two rows further per column.  A block is a pure function of qmf_in_buf
 * (xh_prod2_block, xh_prod3_block, xh_prod4_block); a row's element then adds the blocks that reach it in the order of
 * the columns (xh_prod_gather).  Bands of the three stretch factors are disjoint.
 *
 * Not built: the pitch-adaptive cross products (ixheaacd_hbe_post_anal_xprod2/3/4, :1084-1547: frames with
 * pitch_in_bins * 0.08333333333333 >= 1 are refused), the 4:1 system, the DFT transposer.
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

/* in(row, band): (re, im) of qmf_in_buf[row][2 band ..] */
/* Stretch by 2 (:753-794): the ten products column i adds to rows 1 + 2 i .. of band qb; blk[2 k ..] = product k. */
template <class In>
FX_HD void xh_prod2_block(const In &in, int qb, int i, float *blk) {
  const XhC z0 = in(XH_ZERO_BAND + i, qb);
  const XhC z = xh_norm2(z0.r, z0.i);
  for (int k = 0; k < 10; k++) {
    const XhC n0 = in(1 + i + k, qb);
    const XhC n = xh_norm2(n0.r, n0.i);
    blk[2 * k] = ((n.r * z.r - n.i * z.i) * 0.3333333f);
    blk[2 * k + 1] = ((n.r * z.i + n.i * z.r) * 0.3333333f);
  }
}
/* Stretch by 4 (:1029-1082): six products, rows 3 + 2 i .. */
template <class In>
FX_HD void xh_prod4_block(const In &in, int qb, int i, float *blk) {
  const int inp = qb >> 1, ip = (qb & 1) ? inp + 1 : inp - 1;
  const XhC z0 = in(XH_ZERO_BAND + i, inp);
  const XhC z = xh_norm4(z0.r, z0.i);
  const float temp_r = z.r, temp_i = z.i;
  float zr = z.r, zi = z.i;
  const float temp = zr * zr - zi * zi;
  zi = zr * zi + zi * zr;
  zr = temp_r * temp - temp_i * zi;
  zi = temp_r * zi + temp_i * temp;
  for (int k = 0; k < 6; k++) {
    const XhC n0 = in(i + 2 * k, ip);
    const XhC n = xh_norm4(n0.r, n0.i);
    blk[2 * k] = ((n.r * zr - n.i * zi) * 0.6666667f);
    blk[2 * k + 1] = ((n.r * zi + n.i * zr) * 0.6666667f);
  }
}
/* Stretch by 3 (:796-1027): eight products, rows 2 + 2 i ..; the eight inputs alternate between a sample (rows i + 3 m)
   and a point interpolated from the two rows behind it, normalised by the cube root. */
template <class In>
FX_HD void xh_prod3_block(const In &in, int qb, int i, float *blk) {
  const int inp = (2 * qb) / 3, rem = 2 * qb - 3 * inp;
  float sel[8], sel1[8]; /* ixheaac_sel_case rows (esbr_rom.c:3128) */
  {
    const float t[5][8] = {{1, -1, 1, 1, 1, 1, -1, 1}, {1, 1, -1, 1, 1, -1, 1, 1}, {-1, 1, -1, -1, -1, -1, 1, -1},
                           {-1, -1, 1, -1, -1, 1, -1, -1}, {1, -1, 1, 1, 1, 1, -1, 1}};
    for (int q = 0; q < 8; q++) {
      sel[q] = t[(inp + 1) & 3][q];
      sel1[q] = t[((inp + 1) & 3) + 1][q];
    }
  }
  if (rem == 0 || rem == 1) {
    XhC vec[8];
    for (int m = 0; m < 4; m++) {
      const XhC a = in(i + 3 * m, inp), b = in(i + 3 * m + 2, inp), c = in(i + 3 * m + 1, inp);
      vec[2 * m] = xh_norm3(a.r, a.i);
      float temp_r1 = sel[0] * b.r + sel[1] * b.i;
      float temp_i1 = sel[2] * b.r + sel[3] * b.i;
      temp_r1 += sel[4] * c.r + sel[5] * c.i;
      temp_i1 += sel[6] * c.r + sel[7] * c.i;
      temp_r1 *= 0.3984033437f;
      temp_i1 *= 0.3984033437f;
      vec[2 * m + 1] = xh_norm3(temp_r1, temp_i1);
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
      const XhC a = in(i + 3 * m, inp), a1 = in(i + 3 * m, inp + 1);
      vec[2 * m] = xh_norm3(a1.r, a1.i);
      cap[2 * m] = xh_norm3(a.r, a.i);
      const XhC b = in(i + 3 * m + 2, inp), c = in(i + 3 * m + 1, inp);
      float temp_r1 = sel[0] * b.r + sel[1] * b.i;
      float temp_i1 = sel[2] * b.r + sel[3] * b.i;
      float tmp_cr = temp_r1 + sel[4] * c.r + sel[5] * c.i;
      float tmp_ci = temp_i1 + sel[6] * c.r + sel[7] * c.i;
      const XhC b1 = in(i + 3 * m + 2, inp + 1), c1 = in(i + 3 * m + 1, inp + 1);
      temp_r1 = sel1[0] * b1.r + sel1[1] * b1.i;
      temp_i1 = sel1[2] * b1.r + sel1[3] * b1.i;
      float tmp_vr = temp_r1 + sel1[4] * c1.r + sel1[5] * c1.i;
      float tmp_vi = temp_i1 + sel1[6] * c1.r + sel1[7] * c1.i;
      tmp_cr *= 0.3984033437f;
      tmp_ci *= 0.3984033437f;
      tmp_vr *= 0.3984033437f;
      tmp_vi *= 0.3984033437f;
      vec[2 * m + 1] = xh_norm3(tmp_vr, tmp_vi);
      cap[2 * m + 1] = xh_norm3(tmp_cr, tmp_ci);
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
  for (int i = 0; i < XAAC_HBE_NO_BINS / 2; i++) {
    const int k = r - r0 - 2 * i;
    if (k >= 0 && k < len) acc += blk(i)[2 * k + comp];
  }
  return acc;
}

/* the frame's parameters are usable: bank sizes in the tables, cross-over bands inside the rows, a pitch below the
   cross-product threshold (:1558-1562), the reference's own x_over_qmf[2] > 1 condition (:1573) */
FX_HD bool xh_apply_params_ok(const xaac_hbe_state *st, int pitch_in_bins) {
  const int s = st->synth_size, ks = st->k_start;
  if (!xh_size_ok(s) || ks < 0 || ks + s > 64 || ks * 32 + 2 * s > 7 * 64 || 4 * ks + 4 * s > 128) return false;
  if (st->start_band < 0 || st->end_band > 64 || st->max_stretch < 0 || st->max_stretch > 4) return false;
  if ((float)(pitch_in_bins * 0.08333333333333) >= 1.0f || pitch_in_bins < 0) return false;
  for (int q = 0; q < st->max_stretch && q < 4; q++) {
    if (st->x_over_qmf[q] < 0 || st->x_over_qmf[q] > 64) return false;
    if (q && st->x_over_qmf[q] < st->x_over_qmf[q - 1]) return false;
  }
  if (st->max_stretch >= 4 && st->x_over_qmf[2] <= 1) return false;
  return true;
}

#endif /* XAAC_HBE_TRANS_H */
