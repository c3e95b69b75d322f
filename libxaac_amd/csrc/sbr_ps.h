/*
 * sbr_ps.h -- the fixed-point parametric-stereo decoder of HE-AACv2 for one stream: hybrid analysis of
 * the three lowest QMF bands, transient detection, decorrelation (fractional-delay all-pass chains and
 * plain delays), 2x2 rotation with per-envelope interpolated coefficients, and the delay-line scaling
 * that keeps the PS state in step with the block-floating-point scale of the frame.  One QMF slot at a
 * time, as the reference does inside its synthesis loop (decoder/ixheaacd_qmf_dec.c:1015-1031).
 * Host/device code in the execution model of sbr_core.h: on the GPU one wave runs one stream with the
 * sub-bands / QMF bands / parameter groups on the lanes and the PS state in LDS; the oracle
 * (oracle/oracle_sbr.cpp) runs exactly this source sequentially (lane count 1).
 *
 * Reference map (decoder/...):
 *   xp_filt_2ch / xp_filt_8ch / xp_hybrid_analysis     ixheaacd_hybrid.c:51 / :96 / :214
 *   xp_fft8                                             ixheaacd_dsp_fft32x32s.c:34
 *   xp_divide16_pos                                     ixheaacd_ps_dec.c:212
 *   xp_decorrelation (+ the two all-pass filters)       ixheaacd_ps_dec.c:450, :236, :339
 *   xp_init_rot_env / xp_apply_rot                      ixheaacd_ps_dec.c:714 / :856
 *   xp_ps_headroom / xp_init_ps_scale / xp_scale_states ixheaacd_ps_dec.c:134 / :188, ixheaacd_thumb_ps_dec.c:101
 */
#ifndef XAAC_SBR_PS_H
#define XAAC_SBR_PS_H

#include "fx.h"
#include "../../include/xaac_sbr.h"
#include "sbr_core.h"

#ifndef XP_TABLES_DECLARED
#define XP_TABLES_DECLARED
#if defined(__HIPCC__)
#define XAAC_TAB_QUAL static __device__ const
#include "tables_ps.inc"
#undef XAAC_TAB_QUAL
#else
#include "tables_ps.inc"
#endif
#endif

typedef struct xaac_ps_tables_t XpTables; /* tables_ps.inc: on the GPU the kernel passes its LDS copy */

/* hybrid sub-band samples of one slot: 10 sub-bands (6 of QMF band 0, 2 + 2 of bands 1, 2), left and right */
struct XpHyb {
  int32_t l_re[16], l_im[16], r_re[16], r_im[16];
};

FX_HD int32_t xp_m16x16_shl(int16_t a, int16_t b) { return fx_shl((int32_t)a * b, 1); }

/* hybrid.c:51: 2-channel real-coefficient filter on the 13-slot history w[0..12] */
FX_HD void xp_filt_2ch(const XpTables *T, const int32_t *w_re, const int32_t *w_im, int32_t *h_re, int32_t *h_im) {
  int32_t c_re = 0, c_im = 0;
  for (int t = 0; t < 6; t++) {
    c_re = fx_add_sat(c_re, fx_mul32x16(w_re[1 + 2 * t], T->p2_6[t]));
    c_im = fx_add_sat(c_im, fx_mul32x16(w_im[1 + 2 * t], T->p2_6[t]));
  }
  c_re = fx_shl(c_re, 1);
  c_im = fx_shl(c_im, 1);
  const int32_t m_re = w_re[6] >> 1, m_im = w_im[6] >> 1;
  h_re[0] = fx_add_sat(m_re, c_re);
  h_re[1] = fx_sub_sat(m_re, c_re);
  h_im[0] = fx_add_sat(m_im, c_im);
  h_im[1] = fx_sub_sat(m_im, c_im);
}

/* dsp_fft32x32s.c:34: the 8-point inverse DIT butterfly network that merges to six outputs */
FX_HD void xp_fft8(const int32_t *y, int32_t *re, int32_t *im) {
  int32_t x[16];
  for (int h = 0; h < 2; h++) {
    const int32_t *p = y + 2 * h;
    const int32_t a00 = fx_add_sat(p[0], p[8]), a0 = fx_sub_sat(p[0], p[8]);
    const int32_t a20 = fx_add_sat(p[1], p[9]), a3 = fx_sub_sat(p[1], p[9]);
    const int32_t a10 = fx_add_sat(p[4], p[12]), a2 = fx_sub_sat(p[4], p[12]);
    const int32_t a30 = fx_add_sat(p[5], p[13]), a1 = fx_sub_sat(p[5], p[13]);
    int32_t *o = x + 8 * h;
    o[0] = fx_add_sat(a00, a10);
    o[4] = fx_sub_sat(a00, a10);
    o[1] = fx_add_sat(a20, a30);
    o[5] = fx_sub_sat(a20, a30);
    o[2] = fx_sub_sat(a0, a1);
    o[6] = fx_add_sat(a0, a1);
    o[3] = fx_add_sat(a3, a2);
    o[7] = fx_sub_sat(a3, a2);
  }
  re[0] = fx_add_sat(x[0], x[8]);
  im[0] = fx_add_sat(x[1], x[9]);
  const int32_t a00 = fx_sub_sat(x[0], x[8]), a10 = fx_sub_sat(x[1], x[9]);
  const int32_t a0 = fx_sub_sat(x[4], x[13]), a1 = fx_add_sat(x[5], x[12]);
  re[4] = fx_add_sat(x[4], x[13]);
  im[4] = fx_sub_sat(x[5], x[12]);
  int32_t vr = xs_mul32x16_shl_sat(fx_sub_sat(x[10], x[11]), 0x5A82);
  int32_t vi = xs_mul32x16_shl_sat(fx_add_sat(x[10], x[11]), 0x5A82);
  re[1] = fx_add_sat(x[2], vr);
  im[1] = fx_add_sat(x[3], vi);
  const int32_t a2 = fx_sub_sat(x[2], vr), a3 = fx_sub_sat(x[3], vi);
  re[2] = fx_add_sat(a0, a2);
  im[2] = fx_add_sat(a1, a3);
  vr = xs_mul32x16_shl_sat(fx_add_sat(x[14], x[15]), 0x5A82);
  vi = xs_mul32x16_shl_sat(fx_sub_sat(x[14], x[15]), 0x5A82);
  const int32_t a20 = fx_sub_sat(x[6], vr), a30 = fx_add_sat(x[7], vi);
  re[3] = fx_add_sat(a00, a20);
  im[3] = fx_add_sat(a10, a30);
  re[5] = fx_add_sat(x[6], vr);
  im[5] = fx_sub_sat(x[7], vi);
}

/* hybrid.c:96: 8-channel complex filter (13-tap prototype, symmetric: taps t and t+8 share a phase) */
FX_HD void xp_filt_8ch(const XpTables *T, const int32_t *w_re, const int32_t *w_im, int32_t *h_re, int32_t *h_im) {
  const int16_t tcos = 0x7642, tsin = 0x30fc, tcom = 0x5a82;
  const int16_t *p = T->p8_13;
  int32_t cum[16], re, im;
#define XP_PAIR(t) /* taps t and t + 8 combined */                                                           \
  re = fx_shl(fx_add_sat(fx_mul32x16(w_re[t], p[t]), fx_mul32x16(w_re[(t) + 8], p[(t) + 8])), 1);           \
  im = fx_shl(fx_add_sat(fx_mul32x16(w_im[t], p[t]), fx_mul32x16(w_im[(t) + 8], p[(t) + 8])), 1)
  XP_PAIR(0);
  cum[12] = fx_shl(fx_mul32x16(fx_add_sat(im, re), tcom), 1);
  cum[13] = fx_shl(fx_mul32x16(fx_sub_sat(im, re), tcom), 1);
  XP_PAIR(1);
  cum[10] = fx_shl(fx_add_sat(fx_mul32x16(im, tcos), fx_mul32x16(re, tsin)), 1);
  cum[11] = fx_shl(fx_sub_sat(fx_mul32x16(im, tsin), fx_mul32x16(re, tcos)), 1);
  cum[9] = fx_shl(fx_mul32x16(fx_sub_sat(w_re[2], w_re[10]), p[10]), 1);
  cum[8] = fx_shl(fx_mul32x16(fx_sub_sat(w_im[2], w_im[10]), p[2]), 1);
  XP_PAIR(3);
  cum[6] = fx_shl(fx_sub_sat(fx_mul32x16(im, tcos), fx_mul32x16(re, tsin)), 1);
  cum[7] = fx_shl(fx_neg_sat(fx_add_sat(fx_mul32x16(im, tsin), fx_mul32x16(re, tcos))), 1);
  XP_PAIR(4);
  cum[4] = fx_shl(fx_mul32x16(fx_sub_sat(im, re), tcom), 1);
  cum[5] = fx_shl(fx_mul32x16(fx_neg_sat(fx_add_sat(im, re)), tcom), 1);
#undef XP_PAIR
  re = fx_shl(fx_mul32x16(w_re[5], p[5]), 1);
  im = fx_shl(fx_mul32x16(w_im[5], p[5]), 1);
  cum[2] = fx_shl(fx_sub_sat(fx_mul32x16(re, tcos), fx_mul32x16(im, tsin)), 1);
  cum[3] = fx_shl(fx_add_sat(fx_mul32x16(re, tsin), fx_mul32x16(im, tcos)), 1);
  cum[0] = fx_shl(fx_mul32x16(w_re[6], p[6]), 1);
  cum[1] = fx_shl(fx_mul32x16(w_im[6], p[6]), 1);
  re = fx_shl(fx_mul32x16(w_re[7], p[7]), 1);
  im = fx_shl(fx_mul32x16(w_im[7], p[7]), 1);
  cum[14] = fx_shl(fx_add_sat(fx_mul32x16(im, tsin), fx_mul32x16(re, tcos)), 1);
  cum[15] = fx_shl(fx_sub_sat(fx_mul32x16(im, tcos), fx_mul32x16(re, tsin)), 1);
  xp_fft8(cum, h_re, h_im);
}

/* hybrid.c:214: push one slot of QMF bands 0..2 into the 12-slot histories and filter (one band per lane).
   in_re / in_im: bands 0..2 of the slot six ahead; scale: right shift applied to them (left if negative). */
template <class PS>
FX_HD void xp_hybrid_analysis(const XsCx &cx, const XpTables *T, const int32_t *in_re, const int32_t *in_im, PS *ps, XpHyb *hy,
                              int scale) {
  XS_PAR(band, 0, 3) {
    const int off = band == 0 ? 0 : 4 + 2 * band;
    int32_t w_re[13], w_im[13];
    int32_t *b_re = ps->hyb_buf[band][0], *b_im = ps->hyb_buf[band][1];
    int32_t t_re = in_re[band], t_im = in_im[band];
    if (scale < 0) {
      t_re = fx_shl(t_re, -scale);
      t_im = fx_shl(t_im, -scale);
    } else {
      t_re = fx_shr(t_re, scale);
      t_im = fx_shr(t_im, scale);
    }
    for (int t = 0; t < 12; t++) {
      w_re[t] = b_re[t];
      w_im[t] = b_im[t];
    }
    w_re[12] = t_re;
    w_im[12] = t_im;
    for (int t = 0; t < 11; t++) {
      b_re[t] = w_re[t + 1];
      b_im[t] = w_im[t + 1];
    }
    b_re[11] = t_re;
    b_im[11] = t_im;
    if (band == 0)
      xp_filt_8ch(T, w_re, w_im, &hy->l_re[off], &hy->l_im[off]);
    else
      xp_filt_2ch(T, w_re, w_im, &hy->l_re[off], &hy->l_im[off]);
  }
  cx.sync();
}

/* ps_dec.c:212: sixteen steps of restoring division on the normalised 16-bit heads of op1 <= op2; its only use keeps
   the low half of the result (ps_dec.c:565), which is the quotient floor(u * 2^15 / v) of those heads -- one integer
   divide instead of the bit-serial loop (checked against the loop on 5e7 operand pairs, tools history) */
FX_HD int32_t xp_divide16_pos(int32_t op1, int32_t op2) {
  const int nrm = fx_norm32(op2);
  const uint32_t u = (uint32_t)xs_shl(op1, nrm) >> 16, v = (uint32_t)xs_shl(op2, nrm) >> 16;
  if (u == 0) return 0;
  return (int32_t)(((u << 15) / v) & 0xffffu);
}

FX_HD int32_t xp_power(int32_t re, int32_t im) {
  return fx_add_sat(fx_mul32x16(re, (int16_t)(re >> 16)), fx_mul32x16(im, (int16_t)(im >> 16)));
}

/* one all-pass stage chain (ps_dec.c:262-321 / :377-437): in = delayed sample rotated by the band's
   fractional-delay phase; three serial links with per-link delay lines.  d0: the band's delay-line
   entry {re, im}; new_re16 / new_im16: the band's current sample rounded to 16 bits (what the delay line takes in);
   ser[m]: its entry in link m's buffer at that link's read position. */
FX_HD void xp_allpass(int16_t *d0, int16_t new_re16, int16_t new_im16, const int16_t *phase, int16_t *ser0, int16_t *ser1,
                      int16_t *ser2, const int16_t *ph0, const int16_t *ph1, const int16_t *ph2, int16_t decay0,
                      int16_t decay1, int16_t decay2, int16_t *out_re, int16_t *out_im) {
  const int16_t r0 = d0[0], i0 = d0[1];
  int16_t in_re = (int16_t)(fx_sub_sat((int32_t)r0 * phase[0], (int32_t)i0 * phase[1]) >> 15);
  int16_t in_im = (int16_t)(fx_add_sat((int32_t)r0 * phase[1], (int32_t)i0 * phase[0]) >> 15);
  d0[0] = new_re16; /* round16 of the band's sample, ps_dec.c:268 */
  d0[1] = new_im16;
  int16_t *ser[3] = {ser0, ser1, ser2};
  const int16_t *ph[3] = {ph0, ph1, ph2};
  const int16_t decay[3] = {decay0, decay1, decay2};
  const int16_t s_re[3] = {ser0[0], ser1[0], ser2[0]}, s_im[3] = {ser0[1], ser1[1], ser2[1]}; /* distinct buffers */
  for (int m = 0; m < 3; m++) {
    const int16_t sr = s_re[m], si = s_im[m];
    int16_t t_re = (int16_t)(fx_sub_sat((int32_t)sr * ph[m][0], (int32_t)si * ph[m][1]) >> 15);
    int16_t t_im = (int16_t)(fx_add_sat((int32_t)sr * ph[m][1], (int32_t)si * ph[m][0]) >> 15);
    t_re = (int16_t)(t_re - xs_mult16_shl(in_re, decay[m]));
    t_im = (int16_t)(t_im - xs_mult16_shl(in_im, decay[m]));
    ser[m][0] = (int16_t)(in_re + xs_mult16_shl(t_re, decay[m]));
    ser[m][1] = (int16_t)(in_im + xs_mult16_shl(t_im, decay[m]));
    in_re = t_re;
    in_im = t_im;
  }
  *out_re = in_re;
  *out_im = in_im;
}

/* ps_dec.c:470-545: input power of transient-detector bin `bin` (20 bins: hybrid sub-bands, QMF bands 3..8,
   then groups of QMF bands) */
FX_HD int32_t xp_bin_power(const XpTables *T, int bin, const XpHyb *hy, const int32_t *band_pw, int usb) {
  if (bin < 2) {
    const int a = bin == 0 ? 0 : 4, b = bin == 0 ? 5 : 1;
    int32_t pw = xp_power(hy->l_re[a], hy->l_im[a]);
    pw = fx_add_sat(pw, fx_mul32x16(hy->l_re[b], (int16_t)(hy->l_re[b] >> 16)));
    return fx_add_sat(pw, fx_mul32x16(hy->l_im[b], (int16_t)(hy->l_im[b] >> 16)));
  }
  if (bin < 8) {
    const int sb = T->borders_group[bin + 2];
    return xp_power(hy->l_re[sb], hy->l_im[sb]);
  }
  if (bin < 14) return band_pw[bin - 5];
  const int gr = bin + 2;
  int32_t accu = 0;
  int hi = T->borders_group[gr + 1];
  if (usb < hi) hi = usb;
  const int shift = T->group_shift[gr - 16];
  int sb = T->borders_group[gr];
  for (; sb + 4 <= hi; sb += 4) { /* same order of saturating adds; four loads in flight */
    const int32_t a = band_pw[sb], b = band_pw[sb + 1], c = band_pw[sb + 2], d = band_pw[sb + 3];
    accu = fx_add_sat(fx_add_sat(fx_add_sat(fx_add_sat(accu, a >> shift), b >> shift), c >> shift), d >> shift);
  }
  for (; sb < hi; sb++) accu = fx_add_sat(accu, band_pw[sb] >> shift);
  return accu;
}

/* ps_dec.c:450: decorrelated (right) signal of one slot.  left: the slot's QMF row (64 re | 64 im), right:
   output row; hy: hybrid sub-bands of the slot (left in, right out); ratio: 21 shorts, band_pw: 64 words of
   scratch. */
template <class PS>
FX_HD void xp_decorrelation(const XsCx &cx, const XpTables *T, PS *ps, XpHyb *hy, const int32_t *left, int32_t *right,
                            int16_t *ratio, int32_t *band_pw) {
  const int usb = cx.uni(ps->usb);
  const int idx = cx.uni(ps->idx), idx_long = cx.uni(ps->idx_long);
  const int is0 = cx.uni(ps->idx_ser[0]), is1 = cx.uni(ps->idx_ser[1]), is2 = cx.uni(ps->idx_ser[2]);
  const int32_t *l_re = left, *l_im = left + 64;
  int32_t *r_re = right, *r_im = right + 64;
  XS_PAR(sb, 3, 64) band_pw[sb] = xp_power(l_re[sb], l_im[sb]); /* ps_dec.c:520-545, one QMF band per lane */
  cx.sync();
  XS_PAR(bin, 0, 20) { /* transient detector: peak-decay against smoothed energy, per bin */
    int32_t pw = fx_shl(xp_bin_power(T, bin, hy, band_pw, usb), 1);
    if (pw < 0) pw = 0;
    int32_t pd = fx_mul32x16_shl(ps->peak_decay_diff[bin], 0x620a);
    if (pw > pd) pd = pw;
    ps->peak_decay_diff[bin] = pd;
    int32_t peak_diff = fx_add_sat(fx_mul32x16_shl(ps->peak_decay_diff_prev[bin], 0x6000), fx_sub_sat(pd, pw) >> 2);
    ps->peak_decay_diff_prev[bin] = peak_diff;
    const int32_t nrg = fx_add_sat(fx_mul32x16_shl(ps->energy_prev[bin], 0x6000), pw >> 2);
    ps->energy_prev[bin] = nrg;
    peak_diff = fx_add_sat(peak_diff, peak_diff >> 1);
    ratio[bin] = peak_diff <= nrg ? (int16_t)0x7fff : (int16_t)xp_divide16_pos(nrg, peak_diff);
  }
  XS_ONE ratio[20] = 0;
  cx.sync();
  XS_PAR(u, 0, 10 + 61) { /* u = 0..9: hybrid sub-bands (ps_dec.c:236); above: QMF band u - 7 */
    const int hyb = u < 10, sb = hyb ? u : u - 7;
    if (sb < 23) { /* the thirty all-pass chains (QMF bands 3..22 whatever usb is, ps_dec.c:339): one code path */
      int16_t o_re, o_im;
      const int di = 9 + 3 * (sb - 3);
      int16_t *d0 = hyb ? &ps->sub[idx][2 * sb] : &ps->ap[idx][2 * sb];
      int16_t *e0 = hyb ? &ps->sub_ser[is0][0][2 * sb] : &ps->ser[is0][0][2 * sb];
      int16_t *e1 = hyb ? &ps->sub_ser[is1][1][2 * sb] : &ps->ser[is1][1][2 * sb];
      int16_t *e2 = hyb ? &ps->sub_ser[is2][2][2 * sb] : &ps->ser[is2][2][2 * sb];
      const int16_t *ph = hyb ? &T->frac_delay_phase_fac_qmf_sub_re_im[2 * sb] : &T->frac_delay_phase_fac_qmf_re_im[2 * sb];
      const int16_t *pser = hyb ? &T->frac_delay_phase_fac_qmf_sub_ser_re_im[2 * sb] : &T->frac_delay_phase_fac_qmf_ser_re_im[2 * sb];
      const int pstep = hyb ? 32 : 64;
      xp_allpass(d0, fx_round16(hyb ? hy->l_re[sb] : l_re[sb]), fx_round16(hyb ? hy->l_im[sb] : l_im[sb]), ph, e0, e1, e2, pser, pser + pstep,
                 pser + 2 * pstep, hyb ? T->rev_link_decay_ser[0] : T->decay_scale_factor[di],
                 hyb ? T->rev_link_decay_ser[1] : T->decay_scale_factor[di + 1],
                 hyb ? T->rev_link_decay_ser[2] : T->decay_scale_factor[di + 2], &o_re, &o_im);
      const int16_t tr = ratio[hyb ? T->hybrid_to_bin[sb] : T->delay_to_bin[sb]];
      *(hyb ? &hy->r_re[sb] : &r_re[sb]) = xp_m16x16_shl(o_re, tr);
      *(hyb ? &hy->r_im[sb] : &r_im[sb]) = xp_m16x16_shl(o_im, tr);
    } else if (sb < usb) { /* plain delays: 14 slots for bands 23..34, one slot above (ps_dec.c:602-648) */
      int16_t *d = sb < 35 ? &ps->ld[idx_long][2 * (sb - 23)] : &ps->sd[2 * (sb - 35)];
      const int16_t tr = ratio[sb < 35 ? 18 : 19];
      const int16_t dr = d[0], di = d[1];
      d[0] = fx_round16(l_re[sb]);
      d[1] = fx_round16(l_im[sb]);
      r_re[sb] = xp_m16x16_shl(dr, tr);
      r_im[sb] = xp_m16x16_shl(di, tr);
    }
  }
  cx.sync();
  XS_PAR(sb, usb, 64) { /* after the filters: they run up to band 22 even when usb is lower */
    r_re[sb] = 0;
    r_im[sb] = 0;
  }
  XS_ONE {
    ps->idx_long = (int16_t)(idx_long + 1 >= 14 ? 0 : idx_long + 1);
    ps->idx = (int16_t)(idx + 1 >= 2 ? 0 : idx + 1);
    ps->idx_ser[0] = (int16_t)(is0 + 1 >= ps->sample_ser[0] ? 0 : is0 + 1);
    ps->idx_ser[1] = (int16_t)(is1 + 1 >= ps->sample_ser[1] ? 0 : is1 + 1);
    ps->idx_ser[2] = (int16_t)(is2 + 1 >= ps->sample_ser[2] ? 0 : is2 + 1);
  }
  cx.sync();
}

/* ps_dec.c:678 / :691: quarter-wave table lookups of the rotation angles */
FX_HD int16_t xp_cos512(const XpTables *T, int32_t phi_by_4) {
  const int16_t *trig = T->trig_data;
  int index = fx_round16(fx_abs_sat(phi_by_4)) & 0x3ff;
  return index < 512 ? trig[512 - index] : (int16_t)(-trig[index - 512]);
}
FX_HD int16_t xp_sin512(const XpTables *T, int32_t phi_by_4) {
  const int16_t *trig = T->trig_data;
  int index = fx_round16(phi_by_4);
  if (index < 0) {
    index = (-index) & 0x3ff;
    return index < 512 ? (int16_t)(-trig[index]) : (int16_t)(-trig[1024 - index]);
  }
  index &= 0x3ff;
  return index < 512 ? trig[index] : trig[1024 - index];
}

/* The reference's parser (ixheaacd_ps_bitdec.c:98-300) hands over IID indices within +-7 (+-15 fine), ICC indices
   0..7 and borders 0..32; the tool uses them as raw table indices.  A frame from anywhere else is brought into
   those ranges before use (returns 1 if anything had to be changed: the caller reports status -1 for the stream). */
FX_HD int xp_frame_sanitize(const XsCx &cx, xaac_ps_frame *pf) {
  int bad = 0;
  const int steps = cx.uni(pf->iid_quant) ? 15 : 7;
  XS_PAR(i, 0, (XAAC_PS_MAX_ENV + 2) * XAAC_PS_BANDS_FINE) {
    int16_t *iid = &pf->iid_par_table[0][0] + i, *icc = &pf->icc_par_table[0][0] + i;
    const int a = *iid < -steps ? -steps : (*iid > steps ? steps : *iid);
    const int c = *icc < 0 ? 0 : (*icc > 7 ? 7 : *icc);
    bad |= (a != *iid) | (c != *icc);
    *iid = (int16_t)a;
    *icc = (int16_t)c;
  }
  XS_PAR(i, 0, XAAC_PS_MAX_ENV + 2) {
    const int b = pf->border_position[i];
    const int c = b < 0 ? 0 : (b > 32 ? 32 : b);
    bad |= b != c;
    pf->border_position[i] = (int16_t)c;
  }
  cx.sync();
  return cx.wave_or(bad) != 0;
}

/* ps_dec.c:714: at an envelope border, the target mixing coefficients of every parameter group from the
   IID / ICC indices, and the per-slot increments towards them (one group per lane) */
template <class PS>
FX_HD void xp_rot_env_coeffs(const XsCx &cx, const XpTables *T, PS *ps, const xaac_ps_frame *pf, int env);
template <class PS>
FX_HD void xp_init_rot_env(const XsCx &cx, const XpTables *T, PS *ps, const xaac_ps_frame *pf, int env, int usb) {
  if (env == 0) {
    const int usb_prev = cx.uni(ps->usb);
    cx.sync();
    XS_ONE ps->usb = (int16_t)usb;
    if (usb > usb_prev && usb_prev) { /* clear the delay lines of the bands that just became active */
      const int ap_hi = usb < 23 ? usb : 23;
      if (ap_hi > usb_prev)
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < cx.uni(ps->sample_ser[i]); j++) XS_PAR(k, 2 * usb_prev, 2 * ap_hi) ps->ser[j][i][k] = 0;
      const int ld_hi = usb < 35 ? usb : 35;
      if (ld_hi >= ap_hi && ld_hi <= 12)
        for (int i = 0; i < 14; i++) XS_PAR(k, 2 * ap_hi, 2 * ld_hi) ps->ld[i][k] = 0;
      if (usb >= ld_hi && usb <= 16) XS_PAR(k, 2 * ld_hi, 2 * usb) ps->sd[k] = 0;
    }
    cx.sync();
  }
  xp_rot_env_coeffs(cx, T, ps, pf, env);
}

/* the second half of ps_dec.c:714: target coefficients of envelope `env` and the per-slot increments towards them */
template <class PS>
FX_HD void xp_rot_env_coeffs(const XsCx &cx, const XpTables *T, PS *ps, const xaac_ps_frame *pf, int env) {
  const int fine = cx.uni(pf->iid_quant);
  const int steps = fine ? 15 : 7;
  const int16_t *sf = fine ? T->scale_factors_fine : T->scale_factors;
  int16_t len = fx_sat16((int32_t)cx.uni(pf->border_position[env + 1]) - cx.uni(pf->border_position[env]));
  if (len < 0) len = (int16_t)(len == -32768 ? 32767 : -len);
  if (len > 48) len = 48; /* borders the parser cannot produce (ps_bitdec.c keeps them within 0..32): stay inside the table */
  const int16_t inv_len = xaac_sbr_inv_int_table[len];
  XS_PAR(g, 0, XAAC_PS_GROUPS) {
    const int bin = T->group_to_bin[g];
    const int iid = pf->iid_par_table[env][bin], icc = pf->icc_par_table[env][bin];
    const int16_t c1 = sf[steps + iid], c2 = sf[steps - iid];
    const int32_t beta =
        fx_mul32x16_shl(xp_m16x16_shl(T->alpha_values[icc], (int16_t)(c1 - c2)), 0x5a82);
    const int32_t alpha = xs_shr_dir_sat_limit(xs_shl(T->alpha_values[icc], 16), 1);
    const int16_t bpa = fx_round16(fx_add_sat(beta, alpha)), bma = fx_round16(fx_sub_sat(beta, alpha));
    const int32_t rescale = (int32_t)(0x0517cc1b << 1);
    const int32_t ipa = fx_mul32x16(rescale, bpa), ima = fx_mul32x16(rescale, bma);
    const int16_t h11 = xs_mult16_shl(xp_cos512(T, ipa), c2), h12 = xs_mult16_shl(xp_cos512(T, ima), c1);
    const int16_t h21 = xs_mult16_shl(xp_sin512(T, ipa), c2), h22 = xs_mult16_shl(xp_sin512(T, ima), c1);
    ps->delta_h11_h12[2 * g] = xs_mult16_shl(inv_len, (int16_t)(h11 - ps->h11_h12_vec[2 * g]));
    ps->delta_h11_h12[2 * g + 1] = xs_mult16_shl(inv_len, (int16_t)(h12 - ps->h11_h12_vec[2 * g + 1]));
    ps->delta_h21_h22[2 * g] = xs_mult16_shl(inv_len, (int16_t)(h21 - ps->h21_h22_vec[2 * g]));
    ps->delta_h21_h22[2 * g + 1] = xs_mult16_shl(inv_len, (int16_t)(h22 - ps->h21_h22_vec[2 * g + 1]));
    ps->H11_H12[2 * g] = ps->h11_h12_vec[2 * g];
    ps->H11_H12[2 * g + 1] = ps->h11_h12_vec[2 * g + 1];
    ps->H21_H22[2 * g] = ps->h21_h22_vec[2 * g];
    ps->H21_H22[2 * g + 1] = ps->h21_h22_vec[2 * g + 1];
    ps->h11_h12_vec[2 * g] = h11;
    ps->h11_h12_vec[2 * g + 1] = h12;
    ps->h21_h22_vec[2 * g] = h21;
    ps->h21_h22_vec[2 * g + 1] = h22;
  }
  cx.sync();
}

FX_HD void xp_rotate(int32_t *l, int32_t *r, int16_t h11, int16_t h12, int16_t h21, int16_t h22) {
  const int32_t nl = fx_add_sat(fx_mul32x16(*l, h11), fx_mul32x16(*r, h21));
  const int32_t nr = fx_add_sat(fx_mul32x16(*l, h12), fx_mul32x16(*r, h22));
  *l = fx_shl(nl, 2);
  *r = fx_shl(nr, 2);
}

/* ps_dec.c:856: advance the interpolated coefficients by one slot and mix left / decorrelated into the
   output pair, in the hybrid domain for QMF bands 0..2 (their sub-bands are then summed back) */
template <class PS>
FX_HD void xp_apply_rot(const XsCx &cx, const XpTables *T, PS *ps, XpHyb *hy, int32_t *left, int32_t *right) {
  const int usb = cx.uni(ps->usb);
  XS_PAR(g, 0, XAAC_PS_GROUPS) {
    ps->H11_H12[2 * g] = (int16_t)(ps->H11_H12[2 * g] + ps->delta_h11_h12[2 * g]);
    ps->H11_H12[2 * g + 1] = (int16_t)(ps->H11_H12[2 * g + 1] + ps->delta_h11_h12[2 * g + 1]);
    ps->H21_H22[2 * g] = (int16_t)(ps->H21_H22[2 * g] + ps->delta_h21_h22[2 * g]);
    ps->H21_H22[2 * g + 1] = (int16_t)(ps->H21_H22[2 * g + 1] + ps->delta_h21_h22[2 * g + 1]);
  }
  cx.sync();
  XS_PAR(sb, 0, 10) {
    const int16_t h11 = ps->H11_H12[2 * sb], h12 = ps->H11_H12[2 * sb + 1], h21 = ps->H21_H22[2 * sb],
                  h22 = ps->H21_H22[2 * sb + 1];
    xp_rotate(&hy->l_re[sb], &hy->r_re[sb], h11, h12, h21, h22);
    xp_rotate(&hy->l_im[sb], &hy->r_im[sb], h11, h12, h21, h22);
  }
  cx.sync();
  int32_t *l_re = left, *l_im = left + 64, *r_re = right, *r_im = right + 64;
  XS_PAR(sb, 0, usb > 3 ? usb : 3) {
    if (sb < 3) { /* QMF bands 0..2: the sum of their hybrid sub-bands */
      const int p = sb == 0 ? 0 : 4 + 2 * sb, n = sb == 0 ? 6 : 2;
      int32_t a = hy->l_re[p], b = hy->l_im[p], c = hy->r_re[p], d = hy->r_im[p];
      for (int k = 1; k < n; k++) {
        a = fx_add_sat(a, hy->l_re[p + k]);
        b = fx_add_sat(b, hy->l_im[p + k]);
        c = fx_add_sat(c, hy->r_re[p + k]);
        d = fx_add_sat(d, hy->r_im[p + k]);
      }
      l_re[sb] = a;
      l_im[sb] = b;
      r_re[sb] = c;
      r_im[sb] = d;
    } else { /* the band's parameter group: borders 3,4,...,9,11,14,18,23,35,64 */
      const int g = T->band_to_group[sb];
      const int16_t h11 = ps->H11_H12[2 * g], h12 = ps->H11_H12[2 * g + 1], h21 = ps->H21_H22[2 * g],
                    h22 = ps->H21_H22[2 * g + 1];
      xp_rotate(&l_re[sb], &r_re[sb], h11, h12, h21, h22);
      xp_rotate(&l_im[sb], &r_im[sb], h11, h12, h21, h22);
    }
  }
  cx.sync();
}

/* ---- keeping the PS state in the frame's block-floating-point scale (ps_dec.c:134-210, thumb:101) ---- */
FX_HD int32_t xp_or_abs16(const XsCx &cx, const int16_t *p, int n, int32_t m) {
  XS_PAR(i, 0, n) m |= fx_abs_nrm(p[i]);
  return m;
}
template <class PS>
FX_HD int xp_ps_headroom(const XsCx &cx, const PS *ps) {
  int32_t m = 0;
  for (int i = 0; i < 2; i++) m = xp_or_abs16(cx, &ps->ap[i][6], 40, m);
  m = xp_or_abs16(cx, &ps->ld[0][0], 2 * 14 * 12, m);
  m = xp_or_abs16(cx, ps->sd, 2 * 29, m);
  m = xp_or_abs16(cx, &ps->sub[0][0], 2 * 16 * 2, m);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < cx.uni(ps->sample_ser[i]); j++) m = xp_or_abs16(cx, &ps->ser[j][i][6], 40, m);
  m = xp_or_abs16(cx, &ps->sub_ser[0][0][0], 2 * 3 * 5 * 16, m);
  m = (int32_t)((uint32_t)m << 16);
  const int32_t *h = &ps->hyb_buf[0][0][0];
  XS_PAR(i, 0, 3 * 2 * 12) m |= fx_abs_nrm(h[i]);
  return xs_pnorm32(cx.wave_or(m));
}
FX_HD void xp_scale16(const XsCx &cx, int16_t *p, int n, int scale) { /* scale > 0: left, saturating; < 0: right */
  if (scale > 0) {
    const int s = scale > 15 ? 15 : scale;
    XS_PAR(i, 0, n) p[i] = fx_sat16(xs_shl(p[i], s));
  } else {
    const int s = -scale > 31 ? 31 : -scale;
    XS_PAR(i, 0, n) p[i] = (int16_t)(p[i] >> s);
  }
}
FX_HD void xp_scale32(const XsCx &cx, int32_t *p, int n, int scale) {
  if (scale > 0)
    XS_PAR(i, 0, n) p[i] = fx_shl_sat(p[i], scale);
  else
    XS_PAR(i, 0, n) p[i] = fx_shr(p[i], -scale);
}
template <class PS>
FX_HD void xp_scale_states(const XsCx &cx, PS *ps, int scale) {
  if (scale == 0) return;
  for (int m = 0; m < 2; m++) xp_scale16(cx, &ps->ap[m][6], 40, scale);
  xp_scale16(cx, &ps->ld[0][0], 2 * 14 * 12 + 2 * 29, scale);        /* ld and sd are one block */
  xp_scale16(cx, &ps->sub[0][0], 2 * 16 * 2 + 2 * 3 * 5 * 16, scale); /* sub and sub_ser too */
  for (int i = 0; i < 3; i++)
    for (int m = 0; m < cx.uni(ps->sample_ser[i]); m++) xp_scale16(cx, &ps->ser[m][i][6], 40, scale);
  xp_scale32(cx, &ps->hyb_buf[0][0][0], 2 * 3 * 12, scale);
  xp_scale32(cx, ps->peak_decay_diff, 3 * 20, 2 * scale);
}
/* ps_dec.c:188: returns ps_scale */
template <class PS>
FX_HD int xp_init_ps_scale(const XsCx &cx, PS *ps, int lb_scale, int ov_lb_scale, int hb_scale) {
  const int reserve = xp_ps_headroom(cx, ps);
  const int dbs = (int16_t)(cx.uni(ps->delay_buffer_scale) + reserve);
  int16_t t = (int16_t)(lb_scale < ov_lb_scale ? lb_scale : ov_lb_scale);
  if (hb_scale < t) t = (int16_t)hb_scale;
  if (dbs < t) t = (int16_t)dbs;
  const int ps_scale = t - 1;
  cx.sync();
  xp_scale_states(cx, ps, (int16_t)((ps_scale - dbs) + reserve));
  XS_ONE ps->delay_buffer_scale = (int16_t)ps_scale;
  cx.sync();
  return ps_scale;
}

#endif /* XAAC_SBR_PS_H */
