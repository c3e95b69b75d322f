/*
 * sbr_qmf.h -- per-slot transforms of the SBR QMF banks (fixed-point "Path B"),
 * written once as scalar host/device code: on the GPU one LANE runs one QMF
 * slot (the transforms are 16/32-point FFT sized, there are 32 slots per
 * channel-frame and thousands of channel-frames, so the batch is the parallel
 * axis and no cross-lane exchange is needed); on the host the same code is the
 * arithmetic core of the oracle (oracle/oracle_qmf.cpp), which pins it to the
 * reference function by function.
 *
 * Reference map (decoder/...):
 *   xq_radix4          generic/ixheaacd_qmf_dec_generic.c:1736  ixheaacd_radix4bfly
 *   xq_postradix4/2    generic/...:1831 / :1934                 ixheaacd_postradixcompute4/2
 *   xq_dct3_32         generic/...:63                           ixheaacd_dct3_32        (LP analysis)
 *   xq_cos_sin_mod     generic/...:259                          ixheaacd_cos_sin_mod    (HQ analysis + synthesis)
 *   xq_fwd_modulation  generic/...:468                          ixheaacd_fwd_modulation (HQ analysis)
 *   xq_dct2_64_lp      generic/...:241 + ixheaacd_qmf_dec.c:72-215 (pretwdct2, fftposttw, posttwdct2)
 *                      + generic/...:851 ixheaacd_inv_modulation_lp  (LP synthesis, one slot -> 128 ring samples)
 *   xq_synth_hq_slot   generic/...:869 inv_emodulation + :1638 shiftrountine_with_rnd (HQ synthesis slot)
 * The reference walks these arrays with post-incremented pointers that are also
 * used as array bases; everything below is the same sequential dataflow written
 * with explicit indices.  mul() is (a*b)>>16 with a 16-bit b; "w" ops wrap,
 * "_sat" ops clamp, exactly as in fx.h.
 */
#ifndef XAAC_SBR_QMF_H
#define XAAC_SBR_QMF_H

#include <stdint.h>
#include "fx.h"

#ifndef XQ_TABLES_DECLARED
#define XQ_TABLES_DECLARED
#if defined(__HIPCC__)
#define XAAC_TAB_QUAL static __device__ const
#include "tables_qmf.inc"
#undef XAAC_TAB_QUAL
#define XQ_T(name) xaac_qmf_##name
#else
#include "tables_qmf.inc"
#define XQ_T(name) xaac_qmf_##name
#endif
#endif

#ifndef XQ_ELD_TABLES_DECLARED
#define XQ_ELD_TABLES_DECLARED
#if defined(__HIPCC__)
#define XAAC_TAB_QUAL static __device__ const
#include "tables_qmf_eld.inc"
#undef XAAC_TAB_QUAL
#else
#include "tables_qmf_eld.inc"
#endif
#endif

#ifndef XQ_ESBR_TABLES_DECLARED
#define XQ_ESBR_TABLES_DECLARED
#if defined(__HIPCC__)
#define XAAC_TAB_QUAL static __device__ const
#include "tables_qmf_esbr.inc"
#undef XAAC_TAB_QUAL
#else
#include "tables_qmf_esbr.inc"
#endif
#endif

#if defined(__HIPCC__)
#define XQ_UNROLL _Pragma("unroll")
#else
#define XQ_UNROLL
#endif

FX_HD int32_t xq_mul(int32_t a, int16_t b) { return fx_mul32x16(a, b); }

/* The reference has every transform below twice: with 16-bit twiddles and 32x16 products rounded down one by one (the
   fixed-point "Path B" banks), and with 32-bit twiddles and full 64-bit products summed before the one shift (the
   eSBR "Path A" banks: ixheaacd_esbr_radix4bfly generic:880, ixheaacd_esbr_cos_sin_mod :1163, ...).  The data flow is
   the same, only the two-product primitives differ -- they are the policy W:
     madd / msub   rotation: a wa +- b wb, saturating            (add32_sat of two >>16 products | add64 / sub64_sat, >>32)
     badd / bsub   radix-4 butterfly output, then << 1, wrapping  (generic:1736 | generic:880: RADIXSHIFT = 1)         */
struct XqW16 {
  typedef int16_t T;
  static FX_MEMBER int32_t madd(int32_t a, T wa, int32_t b, T wb) { return fx_add_sat(xq_mul(a, wa), xq_mul(b, wb)); }
  static FX_MEMBER int32_t msub(int32_t a, T wa, int32_t b, T wb) { return fx_sub_sat(xq_mul(a, wa), xq_mul(b, wb)); }
  static FX_MEMBER int32_t badd(int32_t a, T wa, int32_t b, T wb) { return fx_shlw(fx_add(xq_mul(a, wa), xq_mul(b, wb)), 1); }
  static FX_MEMBER int32_t bsub(int32_t a, T wa, int32_t b, T wb) { return fx_shlw(fx_sub(xq_mul(a, wa), xq_mul(b, wb)), 1); }
  static FX_MEMBER const T *w32() { return XQ_T(w_32); }
  static FX_MEMBER const T *w16() { return XQ_T(w_16); }
  template <int M> static FX_MEMBER const T *tw() { return M == 32 ? XQ_T(sin_cos_twiddle_l64) : XQ_T(sin_cos_twiddle_l32); }
  template <int M> static FX_MEMBER const T *alt() { return M == 32 ? XQ_T(alt_sin_twiddle_l64) : XQ_T(alt_sin_twiddle_l32); }
};
FX_HD int64_t xq_sub64_sat(int64_t a, int64_t b) { /* basic_ops40.h:221 */
  const int64_t d = (int64_t)((uint64_t)a - (uint64_t)b);
  if (((a ^ b) & INT64_MIN) != 0 && ((d ^ a) & INT64_MIN) != 0) return a < 0 ? INT64_MIN : INT64_MAX;
  return d;
}
FX_HD int64_t xq_add64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); } /* basic_ops40.h:207 */
struct XqW32 {
  typedef int32_t T;
  static FX_MEMBER int32_t madd(int32_t a, T wa, int32_t b, T wb) { return (int32_t)(xq_add64((int64_t)a * wa, (int64_t)b * wb) >> 32); }
  static FX_MEMBER int32_t msub(int32_t a, T wa, int32_t b, T wb) { return (int32_t)(xq_sub64_sat((int64_t)a * wa, (int64_t)b * wb) >> 32); }
  static FX_MEMBER int32_t badd(int32_t a, T wa, int32_t b, T wb) { return fx_shlw((int32_t)(xq_add64((int64_t)a * wa, (int64_t)b * wb) >> 32), 1); }
  static FX_MEMBER int32_t bsub(int32_t a, T wa, int32_t b, T wb) {
    return fx_shlw((int32_t)((int64_t)((uint64_t)((int64_t)a * wa) - (uint64_t)((int64_t)b * wb)) >> 32), 1);
  }
  static FX_MEMBER const T *w32() { return XQ_T(esbr_w_32); }
  static FX_MEMBER const T *w16() { return XQ_T(esbr_w_16); }
  template <int M> static FX_MEMBER const T *tw() {
    return M == 32 ? XQ_T(esbr_sin_cos_twiddle_l64) : M == 16 ? XQ_T(esbr_sin_cos_twiddle_l32) : M == 12 ? XQ_T(esbr_sin_cos_twiddle_l24) : XQ_T(esbr_sin_cos_twiddle_l16);
  }
  template <int M> static FX_MEMBER const T *alt() {
    return M == 32 ? XQ_T(esbr_alt_sin_twiddle_l64) : M == 16 ? XQ_T(esbr_alt_sin_twiddle_l32) : M == 12 ? XQ_T(esbr_alt_sin_twiddle_l24) : XQ_T(esbr_alt_sin_twiddle_l16);
  }
};

/* radix-4 pass over interleaved complex x; twiddles (si1,co1,si2,co2,si3,co3) per column */
template <class W = XqW16>
FX_HD void xq_radix4(const typename W::T *w, int32_t *x, int index1, int index) {
  const int h2 = 2 * index, l1 = 4 * index, l2 = 6 * index;
  XQ_UNROLL
  for (int b = 0; b < index1; b++) {
    XQ_UNROLL
    for (int i = 0; i < index; i++) {
      int32_t *p = x + 8 * index * b + 2 * i;
      const typename W::T si1 = w[6 * i], co1 = w[6 * i + 1], si2 = w[6 * i + 2], co2 = w[6 * i + 3], si3 = w[6 * i + 4],
                          co3 = w[6 * i + 5];
      int32_t xh0 = fx_add_sat(p[0], p[l1]), xl0 = fx_sub_sat(p[0], p[l1]);
      int32_t xh20 = fx_add_sat(p[h2], p[l2]), xl20 = fx_sub_sat(p[h2], p[l2]);
      int32_t xh1 = fx_add_sat(p[1], p[l1 + 1]), xl1 = fx_sub_sat(p[1], p[l1 + 1]);
      int32_t xh21 = fx_add_sat(p[h2 + 1], p[l2 + 1]), xl21 = fx_sub_sat(p[h2 + 1], p[l2 + 1]);
      int32_t xt0 = fx_sub_sat(xh0, xh20), yt0 = fx_sub_sat(xh1, xh21);
      int32_t xt1 = fx_add_sat(xl0, xl21), xt2 = fx_sub_sat(xl0, xl21);
      int32_t yt2 = fx_add_sat(xl1, xl20), yt1 = fx_sub_sat(xl1, xl20);
      p[0] = fx_add_sat(xh0, xh20);
      p[1] = fx_add_sat(xh1, xh21);
      p[l2] = W::badd(yt2, si3, xt2, co3);
      p[l2 + 1] = W::bsub(yt2, co3, xt2, si3);
      p[l1] = W::badd(yt0, si2, xt0, co2);
      p[l1 + 1] = W::bsub(yt0, co2, xt0, si2);
      p[h2] = W::badd(yt1, si1, xt1, co1);
      p[h2 + 1] = W::bsub(yt1, co1, xt1, si1);
    }
  }
}

/* final radix-4 (twiddle-free) stage + digit-reversed scatter, 16 complex points */
FX_HD void xq_postradix4(int32_t *y, const int32_t *x) {
  XQ_UNROLL
  for (int k = 0; k < 2; k++) {
    const int h = XQ_T(dig_rev_table4_16)[k] >> 2;
    XQ_UNROLL
    for (int half = 0; half < 2; half++) {
      const int32_t *a = x + 16 * k + 8 * half;
      int32_t *o = y + h + 2 * half;
      int32_t xh0 = fx_add_sat(a[0], a[4]), xh1 = fx_add_sat(a[1], a[5]);
      int32_t xl0 = fx_sub_sat(a[0], a[4]), xl1 = fx_sub_sat(a[1], a[5]);
      int32_t yh0 = fx_add_sat(a[2], a[6]), yh1 = fx_add_sat(a[3], a[7]);
      int32_t yl0 = fx_sub_sat(a[2], a[6]), yl1 = fx_sub_sat(a[3], a[7]);
      o[0] = fx_add_sat(xh0, yh0);
      o[1] = fx_add_sat(xh1, yh1);
      o[8] = fx_add_sat(xl0, yl1);
      o[9] = fx_sub_sat(xl1, yl0);
      o[16] = fx_sub_sat(xh0, yh0);
      o[17] = fx_sub_sat(xh1, yh1);
      o[24] = fx_sub_sat(xl0, yl1);
      o[25] = fx_add_sat(xl1, yl0);
    }
  }
}

/* final radix-2 stage + digit-reversed scatter, 32 complex points */
FX_HD void xq_postradix2(int32_t *y, const int32_t *x) {
  XQ_UNROLL
  for (int k = 0; k < 2; k++) {
    XQ_UNROLL
    for (int i = 0; i < 2; i++) {
      const int h = XQ_T(dig_rev_table2_32)[2 * k + i] >> 2;
      XQ_UNROLL
      for (int half = 0; half < 2; half++) {
        const int32_t *a = x + 32 * k + 8 * i + 16 * half;
        int32_t *o = y + h + 2 * half;
        o[0] = fx_add_sat(a[0], a[2]);
        o[1] = fx_add_sat(a[1], a[3]);
        o[32] = fx_sub_sat(a[0], a[2]);
        o[33] = fx_sub_sat(a[1], a[3]);
        o[8] = fx_add_sat(a[4], a[6]);
        o[9] = fx_add_sat(a[5], a[7]);
        o[40] = fx_sub_sat(a[4], a[6]);
        o[41] = fx_sub_sat(a[5], a[7]);
      }
    }
  }
}

/* LP analysis: 64 window-add outputs in[] (clobbered) -> 32 real subband samples out[] */
FX_HD void xq_dct3_32(int32_t *in, int32_t *out) {
  const int16_t *tw = XQ_T(dct23_tw);
  const int16_t *post = XQ_T(post_fft_tbl);
  out[0] = in[48] >> 7;
  out[1] = 0;
  XQ_UNROLL
  for (int n = 1; n < 16; n++) {
    int32_t t0 = fx_add_sat(fx_shr(in[48 + n], 7), fx_shr(in[48 - n], 7));
    int32_t t1 = fx_sub_sat(fx_shr(in[16 + n], 7), fx_shr(in[16 - n], 7));
    int16_t re = tw[4 * n], im = tw[4 * n + 1];
    out[2 * n] = fx_add(xq_mul(t0, re), xq_mul(t1, im));
    out[2 * n + 1] = fx_add(fx_neg(xq_mul(t1, re)), xq_mul(t0, im));
  }
  {
    int16_t re = tw[64], im = tw[65];
    int32_t t = fx_sub_sat(fx_shr(in[32], 7), fx_shr(in[0], 7));
    int32_t c2 = fx_add(xq_mul(t, re), xq_mul(t, im));
    int32_t c3 = fx_add(fx_neg(xq_mul(t, re)), xq_mul(t, im));
    int32_t a0 = out[0], a1 = out[1];
    int32_t u0 = fx_sub(fx_neg(a1), c3), u1 = fx_sub(a0, c2);
    out[0] = fx_add(fx_add(a0, c2), u0) >> 1;
    out[1] = fx_add(fx_sub(a1, c3), u1) >> 1;
  }
  XQ_UNROLL
  for (int n = 1; n <= 8; n++) {
    int32_t a0 = out[2 * n], a1 = out[2 * n + 1], a3 = out[33 - 2 * n], a2 = out[32 - 2 * n];
    int32_t t0 = fx_sub(a0, a2), t1 = fx_add(a0, a2), t2 = fx_add(a1, a3), t3 = fx_sub(a1, a3);
    if (n < 8) {
      int16_t re = post[16 - 2 * n], im = post[2 * n];
      int32_t t4 = fx_add(xq_mul(t0, re), xq_mul(t2, im));
      int32_t t5 = fx_add(fx_neg(xq_mul(t2, re)), xq_mul(t0, im));
      t1 >>= 1;
      t3 >>= 1;
      out[2 * n] = fx_sub(t1, t4);
      out[2 * n + 1] = fx_add(t3, t5);
      out[33 - 2 * n] = fx_add(fx_neg(t3), t5);
      out[32 - 2 * n] = fx_add(t1, t4);
    } else {
      int16_t re = (int16_t)(-post[0]), im = post[16];
      int32_t t4 = fx_sub(xq_mul(t0, re), xq_mul(t2, im));
      int32_t t5 = fx_add(xq_mul(t2, re), xq_mul(t0, im));
      t1 >>= 1;
      t3 >>= 1;
      out[16] = fx_add(t1, t4);
      out[17] = fx_add(t3, t5);
    }
  }
  xq_radix4(XQ_T(w_16), out, 1, 4);
  xq_postradix4(in, out);
  out[0] = in[0];
  out[2] = in[1];
  XQ_UNROLL
  for (int j = 0; j < 7; j++) {
    out[1 + 4 * j] = in[3 + 2 * j];
    out[3 + 4 * j] = in[2 + 2 * j];
    out[30 - 4 * j] = in[19 + 2 * j];
    out[28 - 4 * j] = in[18 + 2 * j];
  }
  out[29] = in[17];
  out[31] = in[16];
}

/* ---- the general FFT's forward transforms of 4, 8 and 12 points (ixheaacd_complex_fft_p2_dec / _p3 with fft_mode = -1,
   decoder/ixheaacd_fft.c:1412 / :2531): what ixheaacd_esbr_cos_sin_mod (generic:1317-1369) calls for the 24- and 16-channel
   analysis banks of 8:3 and 4:1 SBR.  xr / xi: separate real and imaginary words, in place. ------------------------------ */
FX_HD int32_t xq_mul31_sat(int32_t a, int32_t b) { return fx_sat64(((int64_t)a * (int64_t)b) >> 31); } /* fft.c:48 */
/* the radix-4 butterfly of the first pass (fft.c:1476-1501); results in the reference's store order */
FX_HD void xq_fwd_bfly4(int32_t x0r, int32_t x0i, int32_t x1r, int32_t x1i, int32_t x2r, int32_t x2i, int32_t x3r, int32_t x3i, int32_t *y) {
  x0r = fx_add_sat(x0r, x2r);
  x0i = fx_add_sat(x0i, x2i);
  x2r = fx_sub_sat(x0r, fx_shl_sat(x2r, 1));
  x2i = fx_sub_sat(x0i, fx_shl_sat(x2i, 1));
  x1r = fx_add_sat(x1r, x3r);
  x1i = fx_add_sat(x1i, x3i);
  x3r = fx_sub_sat(x1r, fx_shl_sat(x3r, 1));
  x3i = fx_sub_sat(x1i, fx_shl_sat(x3i, 1));
  x0r = fx_add_sat(x0r, x1r);
  x0i = fx_add_sat(x0i, x1i);
  x1r = fx_sub_sat(x0r, fx_shl_sat(x1r, 1));
  x1i = fx_sub_sat(x0i, fx_shl_sat(x1i, 1));
  x2r = fx_add_sat(x2r, x3i);
  x2i = fx_sub_sat(x2i, x3r);
  x3i = fx_sub_sat(x2r, fx_shl_sat(x3i, 1));
  x3r = fx_add_sat(x2i, fx_shl_sat(x3r, 1));
  y[0] = x0r; y[1] = x0i; y[2] = x2r; y[3] = x2i; y[4] = x1r; y[5] = x1i; y[6] = x3i; y[7] = x3r;
}
/* 4 points: input / 8 (C's truncating division, fft.c:1443), one butterfly on the points in natural order (:1452: the digit
   reversal of 0 is 0, the legs are npoints / 2 words apart) */
FX_HD void xq_fft_fwd4(int32_t *xr, int32_t *xi) {
  int32_t y[8];
  xq_fwd_bfly4(xr[0] / 8, xi[0] / 8, xr[1] / 8, xi[1] / 8, xr[2] / 8, xi[2] / 8, xr[3] / 8, xi[3] / 8, y);
  XQ_UNROLL
  for (int i = 0; i < 4; i++) {
    xr[i] = y[2 * i];
    xi[i] = y[2 * i + 1];
  }
}
/* 8 points: input / 8, two butterflies (even points, odd points: :1452-1457 with not_power_4), then the radix-2 stage
   (:1903-1963; del 4, twiddles at node spacing 128 = the table's words 0, 1 and 256, 257).  The reference reports an
   exponent of 4 for it, which its caller applies (generic:1348). */
FX_HD void xq_fft_fwd8(int32_t *xr, int32_t *xi) {
  int32_t y[16];
  xq_fwd_bfly4(xr[0] / 8, xi[0] / 8, xr[2] / 8, xi[2] / 8, xr[4] / 8, xi[4] / 8, xr[6] / 8, xi[6] / 8, y);
  xq_fwd_bfly4(xr[1] / 8, xi[1] / 8, xr[3] / 8, xi[3] / 8, xr[5] / 8, xi[5] / 8, xr[7] / 8, xi[7] / 8, y + 8);
  const int32_t *tw = XQ_T(esbr_fft8_tw);
  XQ_UNROLL
  for (int q = 0; q < 4; q++) { /* complex points q and q + 4 */
    const int32_t w1h = tw[2 * (q & 1)], w1l = tw[2 * (q & 1) + 1];
    const int32_t x0r = y[2 * q], x0i = y[2 * q + 1];
    int32_t x1r = y[2 * q + 8], x1i = y[2 * q + 9], tmp;
    if (q < 2) {
      tmp = fx_sub_sat(xq_mul31_sat(x1r, w1l), xq_mul31_sat(x1i, w1h));
      x1i = fx_add_sat(xq_mul31_sat(x1r, w1h), xq_mul31_sat(x1i, w1l));
    } else {
      tmp = fx_add_sat(xq_mul31_sat(x1r, w1h), xq_mul31_sat(x1i, w1l));
      x1i = fx_sub_sat(xq_mul31_sat(x1i, w1h), xq_mul31_sat(x1r, w1l));
    }
    x1r = tmp;
    y[2 * q + 8] = x0r / 2 - x1r / 2;
    y[2 * q + 9] = x0i / 2 - x1i / 2;
    y[2 * q] = x0r / 2 + x1r / 2;
    y[2 * q + 1] = x0i / 2 + x1i / 2;
  }
  XQ_UNROLL
  for (int i = 0; i < 8; i++) {
    xr[i] = y[2 * i];
    xi[i] = y[2 * i + 1];
  }
}
/* 12 points (fft.c:2531): three 4-point transforms over the points 3 j + i, halving, the two rotations of a group
   (fft_mode < 0: :2586-2607), the 3-point butterfly with sign_dir = -1 (:2493), results to g, 4 + g, 8 + g */
FX_HD void xq_fft_fwd12(int32_t *xr, int32_t *xi) {
  XQ_UNROLL
  for (int i = 0; i < 3; i++) {
    int32_t ar[4], ai[4];
    XQ_UNROLL
    for (int j = 0; j < 4; j++) {
      ar[j] = xr[3 * j + i];
      ai[j] = xi[3 * j + i];
    }
    xq_fft_fwd4(ar, ai);
    XQ_UNROLL
    for (int j = 0; j < 4; j++) {
      xr[3 * j + i] = ar[j];
      xi[3 * j + i] = ai[j];
    }
  }
  const int32_t *wr = XQ_T(esbr_fft12_tw_r), *wi = XQ_T(esbr_fft12_tw_i);
  int32_t yr[12], yi[12];
  XQ_UNROLL
  for (int g = 0; g < 4; g++) {
    int32_t in[6];
    XQ_UNROLL
    for (int q = 0; q < 3; q++) {
      in[2 * q] = xr[3 * g + q] >> 1;
      in[2 * q + 1] = xi[3 * g + q] >> 1;
    }
    XQ_UNROLL
    for (int q = 1; q < 3; q++) {
      const int32_t c = wr[2 * g + q - 1], sn = wi[2 * g + q - 1];
      const int32_t tmp = fx_sub_sat(xq_mul31_sat(in[2 * q], c), xq_mul31_sat(in[2 * q + 1], sn));
      in[2 * q + 1] = fx_add_sat(xq_mul31_sat(in[2 * q], sn), xq_mul31_sat(in[2 * q + 1], c));
      in[2 * q] = tmp;
    }
    const int32_t sinmu = 1859775393; /* -1859775393 * sign_dir */
    const int32_t temp_real = fx_add_sat(in[0], in[2]), temp_imag = fx_add_sat(in[1], in[3]);
    const int32_t add_r = fx_add_sat(in[2], in[4]), add_i = fx_add_sat(in[3], in[5]);
    const int32_t sub_r = fx_sub_sat(in[2], in[4]), sub_i = fx_sub_sat(in[3], in[5]);
    const int32_t p1 = add_r >> 1, p4 = add_i >> 1;
    const int32_t p2 = fx_shlw(fx_mulhi(sub_i, sinmu), 1), p3 = fx_shlw(fx_mulhi(sub_r, sinmu), 1); /* ixheaac_mult32_shl */
    const int32_t temp = fx_sub(in[0], p1);
    yr[g] = fx_add_sat(temp_real, in[4]);
    yi[g] = fx_add_sat(temp_imag, in[5]);
    yr[4 + g] = fx_add_sat(temp, p2);
    yi[4 + g] = fx_sub_sat(fx_sub_sat(in[1], p3), p4);
    yr[8 + g] = fx_sub_sat(temp, p2);
    yi[8 + g] = fx_sub_sat(fx_add_sat(in[1], p3), p4);
  }
  XQ_UNROLL
  for (int i = 0; i < 12; i++) {
    xr[i] = yr[i];
    xi[i] = yi[i];
  }
}

/* complex modulation core shared by HQ analysis (M = 16) and HQ synthesis (M = 32), generic:259.  The reference
   walks a "real" and an "imaginary" half (s[0..2M-1] and s[64..64+2M-1]) side by side through pre-rotation, an
   M-point complex FFT and post-rotation; the halves never meet inside, so each is a function of its own 2M words
   (H = 0: the first half, H = 1: the second; they differ in the signs of the two rotations).  s: the half's 2M
   words in place, t: 2M words of scratch.  The GPU synthesis kernel runs the halves one after the other through a
   half-size LDS tile. */
template <int M, int H, class W = XqW16>
FX_HD void xq_cos_sin_mod_half(int32_t *s, int32_t *t) {
  typedef typename W::T WT;
  const WT *tw = W::template tw<M>();
  const WT *alt = W::template alt<M>();
  XQ_UNROLL
  for (int q = 0; q < M / 2; q++) {
    {
      const WT wim = tw[4 * q], wre = tw[4 * q + 1];
      const int32_t re = s[2 * q], im = s[2 * M - 1 - 2 * q];
      if (H == 0) {
        t[2 * q] = W::madd(re, wre, im, wim);
        t[2 * q + 1] = W::msub(im, wre, re, wim);
      } else {
        t[2 * q] = W::msub(im, wim, re, wre);
        t[2 * q + 1] = W::madd(re, wim, im, wre);
      }
    }
    {
      const WT wim = tw[4 * q + 2], wre = tw[4 * q + 3];
      const int32_t re = s[2 * M - 2 - 2 * q], im = s[2 * q + 1];
      if (H == 0) {
        t[2 * M - 1 - 2 * q] = W::msub(im, wre, re, wim);
        t[2 * M - 2 - 2 * q] = W::madd(re, wre, im, wim);
      } else {
        t[2 * M - 1 - 2 * q] = W::madd(re, wim, im, wre);
        t[2 * M - 2 - 2 * q] = W::msub(im, wim, re, wre);
      }
    }
  }
  if (M == 32) {
    xq_radix4<W>(W::w32(), t, 1, 8);
    xq_radix4<W>(W::w32() + 48, t, 4, 2);
    xq_postradix2(s, t);
  } else if (M == 16) {
    xq_radix4<W>(W::w16(), t, 1, 4);
    xq_postradix4(s, t);
  } else { /* the 24- and 16-channel eSBR analysis banks: the general FFT on separated words (generic:1317-1369) */
    int32_t xr[M], xi[M];
    XQ_UNROLL
    for (int z = 0; z < M; z++) {
      xr[z] = t[2 * z];
      xi[z] = t[2 * z + 1];
    }
    if (M == 12) xq_fft_fwd12(xr, xi);
    else xq_fft_fwd8(xr, xi);
    XQ_UNROLL
    for (int z = 0; z < M; z++) { /* M = 8: << scaleshift (4), a plain C shift */
      s[2 * z] = M == 12 ? xr[z] : fx_shlw(xr[z], 4);
      s[2 * z + 1] = M == 12 ? xi[z] : fx_shlw(xi[z], 4);
    }
  }
  /* post-rotation, in place and order-sensitive (each value is consumed before it is overwritten) */
  int lo = 0, hi = 2 * M - 1, pa = 0;
  WT wim = alt[pa++], wre = alt[pa++];
  if (H == 0) {
    int32_t re = s[hi];
    s[0] = s[0] >> 1;
    lo = 1;
    s[hi] = fx_neg_sat(s[1] >> 1);
    hi--;
    int32_t im = s[hi];
    s[hi--] = W::madd(re, wre, im, wim);
    s[lo++] = W::msub(im, wre, re, wim);
    XQ_UNROLL
    for (int i = 0; i < M / 2 - 1; i++) {
      int32_t im0 = s[lo], re0 = s[lo + 1], re2 = s[hi];
      s[lo++] = W::madd(re0, wim, im0, wre);
      s[hi--] = W::msub(im0, wim, re0, wre);
      wim = alt[pa++];
      wre = alt[pa++];
      im0 = s[hi];
      s[hi--] = W::madd(re2, wre, im0, wim);
      s[lo++] = W::msub(im0, wre, re2, wim);
    }
  } else {
    int32_t re = s[hi];
    s[hi--] = fx_neg_sat(s[lo] >> 1);
    s[lo] = s[lo + 1] >> 1;
    lo++;
    int32_t im = s[hi];
    s[lo++] = fx_neg_sat(W::madd(re, wre, im, wim));
    s[hi--] = W::msub(re, wim, im, wre);
    XQ_UNROLL
    for (int i = 0; i < M / 2 - 1; i++) {
      int32_t im1 = s[lo], re1 = s[lo + 1], re3 = s[hi];
      s[hi--] = fx_neg_sat(W::madd(re1, wim, im1, wre));
      s[lo++] = W::msub(re1, wre, im1, wim);
      wim = alt[pa++];
      wre = alt[pa++];
      im1 = s[hi];
      s[lo++] = fx_neg_sat(W::madd(re3, wre, im1, wim));
      s[hi--] = W::msub(re3, wim, im1, wre);
    }
  }
}

/* both halves: s[0..2M-1] and s[64..64+2M-1] in place; t = 128 words of scratch */
template <int M, class W = XqW16>
FX_HD void xq_cos_sin_mod(int32_t *s, int32_t *t) {
  xq_cos_sin_mod_half<M, 0, W>(s, t);
  xq_cos_sin_mod_half<M, 1, W>(s + 64, t + 64);
}

/* HQ analysis: 64 window-add outputs -> 32 complex subbands, s[0..31] real, s[64..95] imaginary;
   nrot = usb - lsb of the analysis bank (bands that get the final phase rotation) */
/* eld: the LD / ELD bank's post-modulation twiddles (ixheaacd_sbr_t_cos_sin_l32_eld, generic:650-656) */
FX_HD void xq_fwd_modulation(const int32_t *in, int32_t *s, int32_t *t, int nrot, bool eld = false) {
  XQ_UNROLL
  for (int i = 0; i < 32; i++) {
    int32_t a = fx_shr(in[i], 4), b = fx_shr(in[63 - i], 4);
    s[i] = fx_sub_sat(a, b);
    s[64 + i] = fx_add_sat(a, b);
  }
  xq_cos_sin_mod<16>(s, t);
  const int16_t *tc = eld ? xaac_qmf_eld_t_cos_sin_l32 : XQ_T(t_cos_sin_l32);
  XQ_UNROLL
  for (int i = 0; i < 32; i++) {
    if (i < nrot) {
      int32_t re = s[i], im = s[64 + i];
      int16_t c = tc[2 * i], sn = tc[2 * i + 1];
      s[i] = fx_add_sat(fx_mul32x16_shl(re, c), fx_mul32x16_shl(im, sn));
      s[64 + i] = fx_sub_sat(fx_mul32x16_shl(im, c), fx_mul32x16_shl(re, sn));
    }
  }
}

/* LP synthesis slot: 64 real subband samples x[] (clobbered), scratch X[64] ->
   the 128 int16 samples b[0..127] that the reference writes at filter_states + drc_offset */
FX_HD void xq_dct2_64_lp(int32_t *x, int32_t *X, int16_t *b) {
  XQ_UNROLL
  for (int n = 0; n < 32; n++) {
    X[n] = x[2 * n];
    X[63 - n] = x[2 * n + 1];
  }
  xq_radix4(XQ_T(w_32), X, 1, 8);
  xq_radix4(XQ_T(w_32) + 48, X, 4, 2);
  xq_postradix2(x, X);
  {
    const int16_t *post = XQ_T(post_fft_tbl);
    x[0] = fx_shlw(x[0], 1);
    x[1] = fx_shlw(x[1], 1);
    XQ_UNROLL
    for (int k = 1; k <= 16; k++) {
      const int pf = 2 * k, pr = 65 - 2 * k;
      int32_t t0 = x[pf], t1 = x[pf + 1], t3 = x[pr], t2 = x[pr - 1];
      int32_t in2 = fx_sub_sat(t3, t1), in1 = fx_add_sat(t3, t1);
      int32_t d = fx_sub_sat(t0, t2), sm = fx_add_sat(t0, t2);
      int16_t re = post[k], im = post[16 - k];
      int32_t v1 = fx_shlw(fx_sub(xq_mul(in1, re), xq_mul(d, im)), 1);
      int32_t v2 = fx_shlw(fx_add(xq_mul(d, re), xq_mul(in1, im)), 1);
      x[pf] = fx_add_sat(sm, v1);
      x[pf + 1] = fx_add_sat(in2, v2);
      x[pr] = fx_sub_sat(v2, in2);
      x[pr - 1] = fx_sub_sat(sm, v1);
    }
  }
  {
    const int16_t *tw = XQ_T(dct23_tw);
    int16_t *of = b + 32;
    int32_t re0 = x[0], im0 = x[1];
    int32_t half = fx_sat64(((int64_t)re0 + (int64_t)im0) >> 1);
    of[0] = fx_round16(fx_shl(half, 4));
    int32_t last = fx_sub_sat(re0, im0);
    XQ_UNROLL
    for (int n = 1; n < 32; n++) {
      int32_t re = x[2 * n], im = x[2 * n + 1];
      int16_t tr = tw[2 * n], ti = tw[2 * n + 1];
      int32_t o_re = fx_sub_sat(xq_mul(re, tr), xq_mul(im, ti));
      int32_t o_im = fx_add_sat(xq_mul(im, tr), xq_mul(re, ti));
      int16_t r1 = fx_round16(fx_shl(o_re, 4)), i1 = fx_round16(fx_shl(o_im, 4));
      of[n] = r1;
      of[-n] = r1;
      of[64 - n] = i1;
      of[64 + n] = fx_neg16(i1);
    }
    int16_t r1 = fx_round16(fx_shl(xq_mul(last, tw[64]), 4));
    of[32] = r1;
    of[-32] = r1;
    of[64] = 0; /* filter_states[3*M] = 0, generic:863 */
  }
}

/* HQ synthesis slot: s[0..63] real, s[64..127] imaginary (clobbered), t scratch ->
   128 int16 ring samples; shift = out_scale_factor + 1 (qmf_dec.c:1064) */
FX_HD void xq_synth_hq_slot(int32_t *s, int32_t *t, int16_t *b, int shift) {
  xq_cos_sin_mod<32>(s, t);
  XQ_UNROLL
  for (int c = 0; c < 64; c++) {
    b[c] = fx_round16(fx_shl_sat(fx_sub_sat(s[64 + c], s[c]), shift));
    b[64 + c] = fx_round16(fx_shl_sat(fx_add_sat(s[64 + 63 - c], s[63 - c]), shift));
  }
}

/* LD / ELD synthesis slot (qmf_dec.c:1043-1066 with AOT_ER_AAC_ELD): ixheaacd_sbr_pre_twiddle (:788, bands 0..62),
   the 64-channel inverse modulation, ixheaacd_shiftrountine_with_rnd_eld (generic:1672) -> the 128 ring samples b[] */
FX_HD int32_t xq_mul32x16_shl_sat(int32_t a, int16_t b) { /* basic_ops40.h:56 */
  return (a == (int32_t)0x80000000 && b == (int16_t)0x8000) ? (int32_t)0x7fffffff : fx_mul32x16_shl(a, b);
}
FX_HD void xq_synth_eld_slot(int32_t *s, int32_t *t, int16_t *b, int shift) {
  const int16_t *tw = xaac_qmf_eld_synth_cos_sin_l32;
  XQ_UNROLL
  for (int k = 0; k < 63; k++) {
    const int32_t x_re = s[k], x_im = s[64 + k];
    const int16_t c = tw[2 * k], sn = tw[2 * k + 1];
    s[k] = fx_add_sat(fx_mul32x16_shl(x_re, c), xq_mul32x16_shl_sat(x_im, sn));
    s[64 + k] = fx_sub_sat(fx_mul32x16_shl(x_im, c), fx_mul32x16_shl(x_re, sn));
  }
  xq_cos_sin_mod<32>(s, t);
  XQ_UNROLL
  for (int c = 0; c < 64; c++) { /* b[c] <- re - im of band c; b[64 + c] <- -(im + re) of band 63 - c */
    b[c] = fx_round16(fx_shl_sat(fx_sub_sat(s[c], s[64 + c]), shift));
    b[64 + c] = fx_round16(fx_shl_sat(fx_neg_sat(fx_add_sat(s[64 + 63 - c], s[63 - c])), shift));
  }
}

/* ---- down-sampled synthesis bank (32 channels: -dsample / output rates above 48 kHz, sbrdec_initfuncs.c:622,
   :1165) -- the same two slot transforms at half the size ---------------------------------------------------- */

/* LP slot, ixheaacd_dct2_32 (qmf_dec.c:341: pretwdct2_32 :89, radix-4 FFT-16, fftposttw_32 :213, posttwdct2_32 :263):
   32 real subband samples x[] (clobbered), scratch X[32] -> the 64 int16 samples the reference writes at
   filter_states + drc_offset (+ the zero at [3 M], generic:863) */
FX_HD void xq_dct2_32_lp(int32_t *x, int32_t *X, int16_t *b) {
  XQ_UNROLL
  for (int n = 0; n < 16; n++) {
    X[n] = x[2 * n];
    X[31 - n] = x[2 * n + 1];
  }
  xq_radix4(XQ_T(w_16), X, 1, 4);
  xq_postradix4(x, X);
  {
    const int16_t *post = XQ_T(post_fft_tbl);
    x[0] = fx_shlw(x[0], 1);
    x[1] = fx_shlw(x[1], 1);
    XQ_UNROLL
    for (int k = 1; k <= 8; k++) {
      const int pf = 2 * k, pr = 33 - 2 * k;
      int32_t t0 = x[pf], t1 = x[pf + 1], t3 = x[pr], t2 = x[pr - 1];
      int32_t in2 = fx_sub_sat(t3, t1), in1 = fx_add_sat(t3, t1);
      int32_t d = fx_sub_sat(t0, t2), sm = fx_add_sat(t0, t2);
      int16_t re = post[2 * k], im = post[16 - 2 * k];
      int32_t v1 = fx_shlw(fx_sub(xq_mul(in1, re), xq_mul(d, im)), 1);
      int32_t v2 = fx_shlw(fx_add(xq_mul(d, re), xq_mul(in1, im)), 1);
      x[pf] = fx_add_sat(sm, v1);
      x[pf + 1] = fx_add_sat(in2, v2);
      x[pr] = fx_sub_sat(v2, in2);
      x[pr - 1] = fx_sub_sat(sm, v1);
    }
  }
  {
    const int16_t *tw = XQ_T(dct23_tw);
    int16_t *of = b + 16;
    int32_t re0 = x[0], im0 = x[1];
    int32_t half = fx_sat64(((int64_t)re0 + (int64_t)im0) >> 1);
    of[0] = fx_round16(fx_shl_sat(half, 4));
    int32_t last = fx_sub_sat(re0, im0);
    XQ_UNROLL
    for (int n = 1; n < 16; n++) {
      int32_t re = x[2 * n], im = x[2 * n + 1];
      int16_t tr = tw[4 * n], ti = tw[4 * n + 1];
      int32_t o_re = fx_sub_sat(xq_mul(re, tr), xq_mul(im, ti));
      int32_t o_im = fx_add_sat(xq_mul(im, tr), xq_mul(re, ti));
      int16_t r1 = fx_round16(fx_shl_sat(o_re, 4)), i1 = fx_round16(fx_shl_sat(o_im, 4));
      of[n] = r1;
      of[-n] = r1;
      of[32 - n] = i1;
      of[32 + n] = fx_neg16(i1);
    }
    int16_t r1 = fx_round16(fx_shl_sat(xq_mul(last, tw[64]), 4));
    of[16] = r1;
    of[-16] = r1;
    of[32] = 0;
  }
}

/* HQ slot: s[0..31] real, s[64..95] imaginary (clobbered), t scratch -> 64 int16 ring samples
   (inv_emodulation generic:869 with w_16, shiftrountine_with_rnd generic:1638 at len 32) */
FX_HD void xq_synth_hq_slot_ds(int32_t *s, int32_t *t, int16_t *b, int shift) {
  xq_cos_sin_mod<16>(s, t);
  XQ_UNROLL
  for (int c = 0; c < 32; c++) {
    b[c] = fx_round16(fx_shl_sat(fx_sub_sat(s[64 + c], s[c]), shift));
    b[32 + c] = fx_round16(fx_shl_sat(fx_add_sat(s[64 + 31 - c], s[31 - c]), shift));
  }
}

/* ---- eSBR ("Path A", -esbr:1) banks: the same slot transforms on 32-bit constants ------------------------------ */

/* ixheaacd_esbr_fwd_modulation (generic:1463), 32 analysis channels: 64 window-add outputs -> 32 complex subbands,
   s[0..31] real, s[64..95] imaginary.  The final rotation runs over usb - lsb = 32 bands (sbr_dec.c:239). */
FX_HD void xq_esbr_fwd_modulation(const int32_t *in, int32_t *s, int32_t *t) {
  XQ_UNROLL
  for (int i = 0; i < 32; i++) {
    const int32_t a = fx_shr(in[i], 4), b = fx_shr(in[63 - i], 4);
    s[i] = fx_sub_sat(a, b);
    s[64 + i] = fx_add_sat(a, b);
  }
  xq_cos_sin_mod<16, XqW32>(s, t);
  const int32_t *tc = XQ_T(esbr_t_cos_sin_l32);
  XQ_UNROLL
  for (int i = 0; i < 32; i++) {
    const int32_t re = s[i], im = s[64 + i], c = tc[2 * i], sn = tc[2 * i + 1];
    s[i] = (int32_t)(xq_add64((int64_t)re * c, (int64_t)im * sn) >> 31);
    s[64 + i] = (int32_t)(xq_sub64_sat((int64_t)im * c, (int64_t)re * sn) >> 31);
  }
}

/* The 24- and 16-channel analysis banks of 8:3 and 4:1 SBR (sbr_dec.c:213-236; sbrdec_initfuncs.c:766-816).
   NB channels: 2 NB window-add outputs -> NB complex subbands, s[0..NB-1] real, s[64..64+NB-1] imaginary. */
template <int NB>
FX_HD void xq_esbr_fwd_modulation_nb(const int32_t *in, int32_t *s, int32_t *t) {
  XQ_UNROLL
  for (int i = 0; i < NB; i++) {
    const int32_t a = fx_shr(in[i], 4), b = fx_shr(in[2 * NB - 1 - i], 4);
    s[i] = fx_sub_sat(a, b);
    s[64 + i] = fx_add_sat(a, b);
  }
  xq_cos_sin_mod<NB / 2, XqW32>(s, t);
  const int32_t *tc = NB == 32 ? XQ_T(esbr_t_cos_sin_l32) : NB == 24 ? XQ_T(esbr_t_cos_sin_l24) : XQ_T(esbr_t_cos_sin_l16);
  XQ_UNROLL
  for (int i = 0; i < NB; i++) {
    const int32_t re = s[i], im = s[64 + i], c = tc[2 * i], sn = tc[2 * i + 1];
    s[i] = (int32_t)(xq_add64((int64_t)re * c, (int64_t)im * sn) >> 31);
    s[64 + i] = (int32_t)(xq_sub64_sat((int64_t)im * c, (int64_t)re * sn) >> 31);
  }
}
/* what distinguishes the three banks in ixheaacd_esbr_analysis_filt_block / ixheaacd_esbr_qmfanal32_winadd (qmf_dec.c:537-731):
   the prototype's table and the stride it is read with, the step of the window pointers, the output gain */
template <int NB> struct XqEsbrAna {
  static FX_MEMBER const int32_t *win() { return NB == 24 ? XQ_T(esbr_qmf_c_24) : XQ_T(esbr_qmf_c); }
  static constexpr int cs = NB == 32 ? 2 : NB == 24 ? 1 : 4; /* stride of the coefficient reads */
  static constexpr int fo = NB == 24 ? 24 : 64;              /* filt_offset */
  static FX_MEMBER float gain() { return 1.0f / (NB == 32 ? 256.0f : NB == 24 ? 12.0f : 128.0f); }
};
/* one slot's window-add: ring(pos) = the word the reference's anal_filter_states_32 holds at pos when the slot is formed;
   f1 / f2: the two ring offsets (0 and NB, swapped per slot), w1 / w2: the two window offsets; anal: 2 NB words */
template <int NB, class Ring>
FX_HD void xq_esbr_winadd_nb(const Ring &ring, int f1, int f2, int w1, int w2, int32_t *anal) {
  const int32_t *c = XqEsbrAna<NB>::win();
  constexpr int cs = XqEsbrAna<NB>::cs;
  XQ_UNROLL
  for (int n = 0; n < NB; n++) {
    int64_t a1 = 0, a2 = 0;
    XQ_UNROLL
    for (int j = 0; j < 5; j++) {
      a1 = xq_add64(a1, (int64_t)ring(f1 + n + 2 * NB * j) * c[w1 + cs * (n + 2 * NB * j)]);
      a2 = xq_add64(a2, (int64_t)ring(f2 + n + 2 * NB * j) * c[w2 + cs * (n + 2 * NB * j)]);
    }
    anal[n] = (int32_t)(a1 >> 31);
    anal[NB + n] = (int32_t)(a2 >> 31);
  }
}
/* the window pointers' step after a slot (sbr_dec.c:267-277) */
template <int NB>
FX_HD void xq_esbr_win_step(int &w1, int &w2) {
  constexpr int fo = XqEsbrAna<NB>::fo;
  w1 += fo;
  w2 += fo;
  const int tmp = w1;
  w1 = w2;
  w2 = tmp;
  if (w2 > fo * 10) {
    w1 = 0;
    w2 = fo;
  }
}

/* ixheaacd_esbr_inv_modulation (qmf_dec.c:733) + ixheaacd_shiftrountine_with_rnd_hq (generic:1704), 64 synthesis
   channels: s[0..63] real, s[64..127] imaginary (clobbered) -> the slot's 128 WORD32 ring samples */
FX_HD void xq_esbr_synth_slot(int32_t *s, int32_t *t, int32_t *b, int shift) {
  xq_cos_sin_mod<32, XqW32>(s, t);
  XQ_UNROLL
  for (int c = 0; c < 64; c++) {
    b[c] = fx_shl_sat(fx_sub_sat(s[64 + c], s[c]), shift);
    b[64 + c] = fx_shl_sat(fx_add_sat(s[64 + 63 - c], s[63 - c]), shift);
  }
}

/* the same for the down-sampled bank (32 synthesis channels: -dsample / output rates above 48 kHz; sbr_dec.c:556-569, :605-628):
   s[0..31] real, s[64..95] imaginary (clobbered) -> the slot's 64 WORD32 ring samples */
FX_HD void xq_esbr_synth_slot_ds(int32_t *s, int32_t *t, int32_t *b, int shift) {
  xq_cos_sin_mod<16, XqW32>(s, t);
  XQ_UNROLL
  for (int c = 0; c < 32; c++) {
    b[c] = fx_shl_sat(fx_sub_sat(s[64 + c], s[c]), shift);
    b[32 + c] = fx_shl_sat(fx_add_sat(s[64 + 31 - c], s[31 - c]), shift);
  }
}

#endif /* XAAC_SBR_QMF_H */
