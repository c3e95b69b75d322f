/*
 * sbr_core.h -- the serial "control and adjustment" part of the fixed-point SBR decoder for ONE
 * channel-frame, low-power (real-valued) mode: block-floating-point bookkeeping, LPP transposer
 * (HF generation) and envelope adjustment, as scalar host/device code.  On the GPU one LANE runs
 * one channel (channels are independent; the QMF matrix is stored channel-minor so that the
 * lanes' accesses coalesce); on the host the same code is the oracle's arithmetic
 * (oracle/oracle_sbr.cpp), which is pinned to the compiled reference on frames captured from
 * real HE-AAC streams.
 *
 * Reference map (decoder/...):
 *   xs_fix_mant_div / xs_mant_exp_sqrt / xs_fix_div      ixheaacd_basic_funcs.c:66 / :101 / :130
 *   xs_headroom / xs_adjust                               ixheaacd_env_calc.c:1159 / :1099
 *   xs_rescale_x_overlap                                  ixheaacd_sbrdec_lpfuncs.c:453
 *   xs_invfilt_level_emphasis                             ixheaacd_sbrdec_lpfuncs.c:735
 *   xs_covariance_lp / xs_filter1_lp / xs_low_pow_hf_generator   ixheaacd_lpp_tran.c:271 / :665 / :843
 *   xs_map_sineflags                                      ixheaacd_sbrdec_lpfuncs.c:529
 *   xs_energy_per_subband / xs_energy_per_sfb             ixheaacd_env_calc.c:1211 / :1298
 *   xs_subbandgain / xs_calc_subband_gains                ixheaacd_env_calc.c:1382 / :616
 *   xs_avggain / xs_noiselimiting / xs_alias_reduction    ixheaacd_env_calc.c:1454 / :229 / :78
 *   xs_erg_to_amplitude_lp                                ixheaacd_env_calc.c:423
 *   xs_harm_zerotwo_lp / xs_harm_onethree_lp              ixheaacd_env_calc.c:1564 / :1617
 *   xs_adapt_noise_gain / xs_calc_sbrenvelope             ixheaacd_env_calc.c:479 / :692
 *   xs_sbr_core_lp                                        ixheaacd_sbr_dec.c:726-775, :1050-1245, :1283-1308
 * The reference keeps gains/energies as interleaved (mantissa, exponent) WORD16 pairs and
 * truncates through WORD16 assignments in many places; the pairs and every truncation are kept.
 */
#ifndef XAAC_SBR_CORE_H
#define XAAC_SBR_CORE_H

#include "fx.h"
#include "../../include/xaac_sbr.h"

#ifndef XS_TABLES_DECLARED
#define XS_TABLES_DECLARED
#if defined(__HIPCC__)
#define XAAC_TAB_QUAL static __device__ const
#include "tables_sbr.inc"
#undef XAAC_TAB_QUAL
#else
#include "tables_sbr.inc"
#endif
#endif

#define XS_MAXF XAAC_SBR_MAX_FREQ_COEFFS

/* x86-style shifts (count taken mod 32), which is what the reference's plain << and >> on int compile to */
FX_HD int32_t xs_shl(int32_t v, int s) { return (int32_t)((uint32_t)v << (s & 31)); }
FX_HD int32_t xs_sar(int32_t v, int s) { return v >> (s & 31); }
FX_HD int xs_pnorm32(int32_t a) { return fx_norm32(a); } /* non-negative arguments only */
FX_HD int16_t xs_mult16_shl_sat(int16_t a, int16_t b) { return fx_sat16(((int32_t)a * b) >> 15); }
FX_HD int16_t xs_mult16_shl(int16_t a, int16_t b) { return (int16_t)(((int32_t)a * b) >> 15); }
FX_HD int32_t xs_mult16x16_shl(int16_t a, int16_t b) { return fx_shl((int32_t)a * b, 1); }
FX_HD int32_t xs_mac16x16_shl_sat(int32_t acc, int16_t b, int16_t c) {
  int32_t p = (int32_t)b * c;
  p = (p != (int32_t)0x40000000) ? fx_shl(p, 1) : FX_MAX32;
  return fx_add_sat(acc, p);
}
FX_HD int16_t xs_shl16_sat(int16_t a, int s) {
  if (s > 15) s = 15;
  return fx_sat16(xs_shl(a, s));
}
/* mult32x16hin32: a * (b >> 16) >> 16 */
FX_HD int32_t xs_mul_hi16(int32_t a, int32_t b) { return fx_mul32x16(a, (int16_t)(b >> 16)); }
FX_HD int32_t xs_mul32x16_shl_sat(int32_t a, int16_t b) {
  if (a == FX_MIN32 && b == (int16_t)-32768) return FX_MAX32;
  return fx_mul32x16_shl(a, b);
}
/* basic_ops.h:100-112 shr32_dir_sat_limit */
FX_HD int32_t xs_shr_dir_sat_limit(int32_t a, int b) {
  if (b < 0) return fx_shl_sat(a, -b);
  return fx_shr(a, b > 31 ? 31 : b);
}

/* ---- pseudo-float helpers (ixheaacd_basic_funcs.c) ------------------------------------------- */
FX_HD int xs_fix_mant_div(int16_t op1, int16_t op2, int16_t *res) {
  int pre = fx_norm32(op2) - 16, post;
  int idx = xs_sar(xs_shl(op2, pre), 16 - 3 - 8) & 511;
  if (idx == 0) {
    post = fx_norm32(op1) - 16;
    *res = (int16_t)xs_shl(op1, post);
  } else {
    idx = (idx - 1) >> 1;
    int32_t ratio = (int32_t)xaac_sbr_inv_table[idx] * op1;
    post = fx_norm32(ratio) - 1;
    *res = (int16_t)(xs_shl(ratio, post) >> 15);
  }
  return pre - post;
}

FX_HD void xs_mant_exp_sqrt(int16_t *me) {
  int32_t m = me[0], e = me[1], rm, re;
  if (m > 0) {
    int pre = fx_norm32((int16_t)m) - 16;
    e -= pre;
    int idx = xs_sar(xs_shl(m, pre), 16 - 3 - 8) & 511;
    rm = xaac_sbr_sqrt_table[idx >> 1];
    if (e & 1) {
      rm = (rm * 0x5a82) >> 16;
      e += 3;
    }
    re = e >> 1;
  } else {
    rm = 0;
    re = -16;
  }
  me[0] = (int16_t)rm;
  me[1] = (int16_t)re;
}

FX_HD int32_t xs_fix_div(int32_t op1, int32_t op2) {
  int32_t q = 0;
  int32_t n1 = op1 >> 1, d1 = op2 >> 1;
  uint32_t num = (uint32_t)(n1 < 0 ? -n1 : n1), den = (uint32_t)(d1 < 0 ? -d1 : d1);
  if (num != 0) {
    for (int k = 15; k > 0; k--) {
      q <<= 1;
      num <<= 1;
      if (num >= den) {
        num -= den;
        q++;
      }
    }
  }
  return ((op1 ^ op2) < 0) ? -q : q;
}

/* accumulate (m, e) into a running (am, ae) pseudo-float sum: the recurring idiom of env_calc.c */
FX_HD void xs_acc_me(int32_t *am, int32_t *ae, int32_t m, int32_t e) {
  int32_t d = e - *ae;
  if (d >= 0) {
    *am = m + fx_shr(*am, d);
    *ae = e;
  } else {
    *am = fx_shr(m, -d) + *am;
  }
}

/* ---- QMF matrix view: slot rows of 64 bands; rows -2,-1 are the LPC history --------------------- */
template <class T>
struct XsMat {
  T *p;       /* element (slot, band) at p[((slot + 2) * 64 + band) * stride] */
  int stride;
  FX_MEMBER T &operator()(int slot, int band) const { return p[((slot + 2) * 64 + band) * stride]; }
};
typedef XsMat<int32_t> XsQmf;

/* env_calc.c:1159 (real-valued): headroom of bands [b0,b1) x slots [s0,s1) */
FX_HD int xs_headroom(const XsQmf &x, int b0, int b1, int s0, int s1) {
  int32_t m = 1;
  for (int l = s0; l < s1; l++)
    for (int k = b0; k < b1; k++) m |= fx_abs_nrm(x(l, k));
  return xs_pnorm32(m);
}
/* env_calc.c:1099 (real-valued) */
FX_HD void xs_adjust(const XsQmf &x, int b0, int b1, int s0, int s1, int shift) {
  if (shift == 0) return;
  if (shift > 31) shift = 31;
  if (shift < -31) shift = -31;
  for (int l = s0; l < s1; l++)
    for (int k = b0; k < b1; k++) x(l, k) = shift > 0 ? fx_shlw(x(l, k), shift) : (x(l, k) >> -shift);
}

/* ---- HF generator, low-power mode ------------------------------------------------------------------ */
struct XsCov {
  int32_t phi_11, phi_22, phi_01, phi_02, phi_12, d;
};

/* lpp_tran.c:271: real autocorrelation of band k over `len` (= 38) slots starting at row -2 */
FX_HD void xs_covariance_lp(const XsQmf &x, int k, int len, XsCov *c) {
  int32_t p01 = 0, p02 = 0, p11 = 0;
  int row = -2;
  int32_t t1 = fx_shr(x(row, k), 3), t2 = fx_shr(x(row + 1, k), 3), t3 = 0;
  row += 2;
  /* the reference walks three samples at a time; the running sums only depend on the sample order */
  int n = len; /* number of lags accumulated: slots 2 .. len+1 relative to the start */
  for (int j = 0; j < n; j++) {
    t3 = fx_shr(x(row++, k), 3);
    p01 = fx_add(p01, xs_mul_hi16(t3, t2));
    p02 = fx_add(p02, xs_mul_hi16(t3, t1));
    p11 = fx_add(p11, xs_mul_hi16(t2, t2));
    t1 = t2;
    t2 = t3;
  }
  /* after the loop (t1, t2) are the last two samples; the reference's temp1/temp3 at that point */
  int32_t last1 = t2, last3 = t1;
  int32_t first = fx_shr(x(-2, k), 3), second = fx_shr(x(-1, k), 3);
  int32_t p12 = fx_add(fx_sub(p01, xs_mul_hi16(last1, last3)), xs_mul_hi16(second, first));
  int32_t p22 = fx_add(fx_sub(p11, xs_mul_hi16(last3, last3)), xs_mul_hi16(first, first));
  int32_t mx = fx_abs_nrm(p01) | fx_abs_nrm(p02) | fx_abs_nrm(p12) | p11 | p22;
  int q = xs_pnorm32(mx);
  c->phi_11 = xs_shl(p11, q);
  c->phi_22 = xs_shl(p22, q);
  c->phi_01 = xs_shl(p01, q);
  c->phi_02 = xs_shl(p02, q);
  c->phi_12 = xs_shl(p12, q);
  c->d = fx_sub_sat(fx_mul32(c->phi_22, c->phi_11), fx_mul32(c->phi_12, c->phi_12));
}

/* sbrdec_lpfuncs.c:735 */
FX_HD void xs_invfilt_level_emphasis(const int32_t *bw_prev, int n, const int32_t *mode, const int32_t *mode_prev,
                                     int32_t *bw) {
  for (int i = 0; i < n; i++) {
    int32_t b = xaac_sbr_new_bw_table[4 * mode_prev[i] + mode[i]];
    int16_t w1, w2;
    if (b < bw_prev[i]) {
      w1 = 0x6000;
      w2 = 0x2000;
    } else {
      w1 = 0x7400;
      w2 = 0x0c00;
    }
    int32_t a = fx_add(fx_mul32x16_shl(b, w1), fx_mul32x16_shl(bw_prev[i], w2));
    if (a < 0x02000000) a = 0;
    if (a >= 0x7f800000) a = 0x7f800000;
    bw[i] = a;
  }
}

/* lpp_tran.c:629 + :665: LPC coefficients per low band, aliasing degrees, and the patch copy/filter.
   start/stop: first_slot_offset and (num_columns + last_slot_offset) as in lpp_tran.c:861 */
FX_HD void xs_filter1_lp(const xaac_sbr_header *h, const XsQmf &x, const XsCov *cov, const int32_t *bw_array,
                         int16_t *degree_alias, int start_idx, int stop_idx, int max_qmf_subband, int start_patch,
                         int stop_patch) {
  int16_t k1, k1_below = 0, k1_below2 = 0;
  int bw_index[XAAC_SBR_MAX_PATCHES] = {0, 0, 0, 0, 0, 0};
  for (int lb = start_patch; lb < stop_patch; lb++) {
    const XsCov *c = &cov[lb];
    int16_t alpha0 = 0, alpha1 = 0;
    if (c->d != 0) {
      int norm_d = fx_norm32(c->d);
      int16_t inv_d = (int16_t)xs_fix_div(0x40000000, xs_shl(c->d, norm_d));
      int32_t mod_d = c->d < 0 ? -c->d : c->d;
      int32_t t = fx_sub_sat(fx_mul32(c->phi_01, c->phi_12), fx_mul32(c->phi_02, c->phi_11)) >> 2;
      if ((t < 0 ? -t : t) < mod_d) alpha1 = (int16_t)(xs_shl(xs_mul32x16_shl_sat(t, inv_d), norm_d) >> 15);
      t = fx_sub_sat(fx_mul32(c->phi_02, c->phi_12), fx_mul32(c->phi_01, c->phi_22)) >> 2;
      if ((t < 0 ? -t : t) < mod_d) alpha0 = (int16_t)(xs_shl(xs_mul32x16_shl_sat(t, inv_d), norm_d) >> 15);
    }
    if (c->phi_11 == 0) {
      k1 = 0;
    } else if (fx_abs_sat(c->phi_01) >= c->phi_11) {
      k1 = c->phi_01 < 0 ? (int16_t)0x7fff : (int16_t)-0x8000;
    } else {
      k1 = (int16_t)(-((int16_t)xs_fix_div(c->phi_01, c->phi_11)));
    }
    if (lb > 1) {
      int16_t deg = fx_sat16(0x7fff - (int32_t)xs_mult16_shl_sat(k1_below, k1_below));
      degree_alias[lb] = 0;
      if (((lb & 1) == 0) && (k1 < 0)) {
        if (k1_below < 0) {
          degree_alias[lb] = 0x7fff;
          if (k1_below2 > 0) degree_alias[lb - 1] = deg;
        } else if (k1_below2 > 0) {
          degree_alias[lb] = deg;
        }
      }
      if (((lb & 1) != 0) && (k1 > 0)) {
        if (k1_below > 0) {
          degree_alias[lb] = 0x7fff;
          if (k1_below2 < 0) degree_alias[lb - 1] = deg;
        } else if (k1_below2 < 0) {
          degree_alias[lb] = deg;
        }
      }
    }
    k1_below2 = k1_below;
    k1_below = k1;

    for (int patch = 0; patch < h->num_patches; patch++) {
      const xaac_sbr_patch *pp = &h->patch[patch];
      int hb = xs_shl(lb + pp->dst_end_band, 8) >> 8;
      if (lb < pp->src_start_band || lb >= pp->src_end_band || hb < max_qmf_subband) continue;
      int bi = bw_index[patch];
      while (hb >= h->bw_borders[bi]) bi++;
      bw_index[patch] = bi;
      int16_t bw = (int16_t)(bw_array[bi] >> 16);
      int32_t a0 = xs_mult16x16_shl(bw, alpha0);
      bw = xs_mult16_shl_sat(bw, bw);
      int32_t a1 = xs_mult16x16_shl(bw, alpha1);
      const int len = stop_idx - start_idx - 1;
      if (bw > 0) {
        /* lpp_tran.c:629: second-order FIR on the low band, two slots per step */
        int r = start_idx - 2; /* row of sub_sig_x[start_idx] in slot coordinates */
        int32_t prev2 = x(r, lb), prev1 = x(r + 1, lb);
        int rl = r + 2, rh = start_idx;
        for (int i = len; i >= 0; i -= 2) {
          int32_t curr = x(rl++, lb);
          int32_t t = xs_mul_hi16(prev2, a1);
          x(rh++, hb) = fx_add_sat(curr >> 2, fx_shlw(fx_add(t, xs_mul_hi16(prev1, a0)), 1));
          prev2 = x(rl++, lb);
          t = xs_mul_hi16(prev1, a1);
          x(rh++, hb) = fx_add_sat(prev2 >> 2, fx_shlw(fx_add(t, xs_mul_hi16(curr, a0)), 1));
          prev1 = prev2;
          prev2 = curr;
        }
      } else {
        for (int i = 0; i <= len; i++) x(start_idx + i, hb) = x(start_idx + i, lb) >> 2;
      }
    }
  }
}

/* lpp_tran.c:843.  degree_alias[64] must be zeroed by the caller.  Writes bw_array_prev. */
FX_HD void xs_low_pow_hf_generator(const xaac_sbr_header *h, xaac_sbr_state *st, const XsQmf &x, int16_t *degree_alias,
                                   int start_idx, int last_slot_offset, int max_qmf_subband, const int32_t *invf_mode,
                                   const int32_t *invf_mode_prev, int norm_max) {
  int32_t bw_array[XAAC_SBR_MAX_PATCHES] = {0, 0, 0, 0, 0, 0};
  XsCov cov[32];
  const int num_patches = h->num_patches;
  const int auto_corr_len = h->num_columns + 6;
  const int stop_idx = h->num_columns + last_slot_offset;
  xs_invfilt_level_emphasis(st->bw_array_prev, h->num_if_bands, invf_mode, invf_mode_prev, bw_array);
  const int actual_stop = (int16_t)(h->patch[num_patches - 1].dst_start_band + h->patch[num_patches - 1].num_bands_in_patch);
  {
    int len = 6;
    if (len > stop_idx) len = stop_idx;
    for (int l = start_idx; l <= len - 1; l++)
      for (int k = actual_stop; k < 64; k++) x(l, k) = 0;
    if (actual_stop < 32)
      for (int l = len; l <= stop_idx - 1; l++)
        for (int k = actual_stop; k < 32; k++) x(l, k) = 0;
  }
  int start_patch = h->start_patch - 2;
  if (start_patch < 1) start_patch = 1;
  int stop_patch = h->patch[0].dst_start_band;
  for (int i = 0; i < 2; i++)
    for (int k = 0; k < stop_patch; k++) x(i - 2, k) = st->lpc_real[i][k];
  for (int k = 0; k < 32; k++) cov[k] = XsCov{0, 0, 0, 0, 0, 0};
  if (norm_max != 30)
    for (int k = start_patch; k < stop_patch; k++) xs_covariance_lp(x, k, auto_corr_len, &cov[k]);
  xs_filter1_lp(h, x, cov, bw_array, degree_alias, start_idx, stop_idx, max_qmf_subband, start_patch, stop_patch);
  for (int lb = h->start_patch; lb < h->stop_patch; lb++) {
    for (int patch = 0; patch < num_patches; patch++) {
      const xaac_sbr_patch *pp = &h->patch[patch];
      int hb = lb + pp->dst_end_band;
      if (lb < pp->src_start_band || lb >= pp->src_end_band || hb >= 64) continue;
      if (hb != pp->dst_start_band) degree_alias[hb] = degree_alias[lb];
    }
  }
  for (int i = 0; i < h->num_if_bands; i++) st->bw_array_prev[i] = bw_array[i];
}

/* ---- envelope adjuster ------------------------------------------------------------------------------ */
/* sbrdec_lpfuncs.c:529 */
FX_HD void xs_map_sineflags(const int16_t *tbl_hi, int nsf, const uint8_t *add_harm, int8_t *flags_prev, int tr_env,
                            int8_t *sine_mapped) {
  const int low2 = tbl_hi[0] << 1;
  for (int i = 0; i < XS_MAXF; i++) sine_mapped[i] = XAAC_SBR_MAX_ENVELOPES;
  int8_t *fp = flags_prev;
  for (int i = nsf - 1; i >= 0; i--) {
    int old = *fp;
    *fp++ = (int8_t)add_harm[i];
    if (add_harm[i]) {
      int q = ((tbl_hi[i + 1] + tbl_hi[i]) - low2) >> 1;
      sine_mapped[q] = old ? 0 : (int8_t)tr_env;
    }
  }
}

/* env_calc.c:1211, low-power branch: per-band energy estimate over slots [s0,s1) */
FX_HD void xs_energy_per_subband(const XsQmf &x, int s0, int s1, int b0, int b1, int frame_exp, int16_t *nrg_est) {
  const int16_t inv_width = xaac_sbr_inv_int_table[s1 - s0];
  const int n = s1 - s0;
  frame_exp <<= 1;
  for (int k = b0; k < b1; k++) {
    int32_t mx = 1;
    for (int l = 0; l < n; l++) {
      int32_t v = fx_abs_nrm(x(s0 + l, k));
      if (v > mx) mx = v;
    }
    int pre = xs_pnorm32(mx) - 3;
    int32_t accu = 0;
    int shift = 16 - pre;
    for (int l = 0; l < n; l++) {
      int16_t t = shift > 0 ? (int16_t)xs_sar(x(s0 + l, k), shift) : (int16_t)xs_shl(x(s0 + l, k), -shift);
      accu = fx_add(accu, (int32_t)t * t);
    }
    if (accu != 0) {
      shift = -xs_pnorm32(accu);
      int16_t sum_m = (int16_t)xs_shr_dir_sat_limit(accu, 16 + shift);
      *nrg_est++ = xs_mult16_shl_sat(sum_m, inv_width);
      shift = shift - (pre << 1) + 1;
      *nrg_est++ = (int16_t)(frame_exp + shift + 1);
    } else {
      *nrg_est++ = 0;
      *nrg_est++ = 0;
    }
  }
}

/* env_calc.c:1298, low-power branch */
FX_HD void xs_energy_per_sfb(const XsQmf &x, int nsf, const int16_t *tbl, int s0, int s1, int max_sb, int frame_exp,
                             int16_t *nrg_est) {
  const int16_t inv_width = xaac_sbr_inv_int_table[s1 - s0];
  frame_exp <<= 1;
  for (int j = 0; j < nsf; j++) {
    int li = tbl[j];
    if (li < max_sb) continue;
    int ui = tbl[j + 1];
    int pre = xs_headroom(x, li, ui, s0, s1) - 4;
    int32_t accumulate = 0;
    for (int k = li; k < ui; k++) {
      int p1 = 16 - pre;
      if (p1 > 31) p1 = 31;
      int32_t line = 0;
      for (int l = s0; l < s1; l++) {
        int16_t t = (int16_t)fx_shr_dir(x(l, k), p1);
        line = fx_add_sat(line, (int32_t)t * t);
      }
      accumulate = fx_add_sat(accumulate, fx_shr(line, 9));
    }
    int shift = xs_pnorm32(accumulate);
    int16_t sum_m = (int16_t)xs_shr_dir_sat_limit(accumulate, 16 - shift);
    int32_t sum_e;
    if (sum_m == 0) {
      sum_e = 0;
    } else {
      sum_m = xs_mult16_shl_sat(sum_m, inv_width);
      sum_m = xs_mult16_shl_sat(sum_m, xaac_sbr_inv_int_table[ui - li]);
      sum_e = ((frame_exp + 11) - shift) - (pre << 1);
    }
    for (int k = li; k < ui; k++) {
      *nrg_est++ = sum_m;
      *nrg_est++ = (int16_t)sum_e;
    }
  }
}

/* env_calc.c:1382 */
FX_HD void xs_subbandgain(int16_t e_orig_m, int16_t noise_m, int16_t est_m, int16_t est_e, int16_t noise_e,
                          int16_t ref_e, int sine_present, int sine_mapped, int noise_absc, int16_t *gain,
                          int16_t *noise_floor, int16_t *sine) {
  int16_t v1m, v1e, v2m, v2e, v3m, v3e;
  if (est_m == 0) {
    est_m = 0x4000;
    est_e = 1;
  }
  v1m = xs_mult16_shl_sat(e_orig_m, noise_m);
  v1e = (int16_t)(ref_e + noise_e);
  {
    int32_t accu, d = noise_e - 1;
    if (d >= 0) {
      accu = noise_m + fx_shr(0x4000, d);
      v2e = noise_e;
    } else {
      accu = fx_shr((int32_t)noise_m, -d) + 0x4000;
      v2e = 1;
    }
    if ((accu < 0 ? -accu : accu) >= 0x8000) {
      accu >>= 1;
      v2e++;
    }
    v2m = (int16_t)accu;
  }
  int t = xs_fix_mant_div(v1m, v2m, noise_floor);
  noise_floor[1] = (int16_t)(t + (v1e - v2e) + 1);
  if (sine_present || !noise_absc) {
    v3m = xs_mult16_shl_sat(v2m, est_m);
    v3e = (int16_t)(v2e + est_e);
  } else {
    v3m = est_m;
    v3e = est_e;
  }
  if (!sine_present) {
    v1m = e_orig_m;
    v1e = ref_e;
  }
  t = xs_fix_mant_div(v1m, v3m, gain);
  gain[1] = (int16_t)(t + (v1e - v3e) + 1);
  if (sine_present && sine_mapped) {
    t = xs_fix_mant_div(e_orig_m, v2m, sine);
    sine[1] = (int16_t)(t + (ref_e - v2e) + 1);
  }
}

/* env_calc.c:616 */
FX_HD void xs_calc_subband_gains(const xaac_sbr_header *h, const xaac_sbr_frame *f, int freq_res,
                                 const int16_t *noise_floor, int nsf, int mvalue, int env, const int8_t *sine_mapped,
                                 int8_t *alias_red, int16_t *e_orig, int16_t *sine, const int16_t *est, int16_t *gain,
                                 int16_t *noise_lvl, int noise_absc) {
  const int16_t *tbl = freq_res ? h->freq_band_tbl_hi : h->freq_band_tbl_lo;
  int ui_noise = h->freq_band_tbl_noise[1], nb = 0, c = 0;
  const int sb_start = h->sub_band_start;
  const int skip = f->max_qmf_subband_aac - sb_start;
  const int8_t *sm = sine_mapped;
  const int8_t *sm1 = sine_mapped + skip;
  const int16_t *env_sf = &f->int_env_sf_arr[mvalue];
  int8_t *ar = &alias_red[tbl[0] - sb_start];
  int16_t nm = (int16_t)(noise_floor[nb] & 0xffc0), ne = (int16_t)((noise_floor[nb] & 63) - 38);
  for (int j = 0; j < nsf; j++) {
    int li = tbl[j], ui = tbl[j + 1];
    int16_t sf = *env_sf++;
    int16_t ref_e = (int16_t)((sf & 63) - 16), ref_m = (int16_t)(sf & 0xffc0);
    int present = 0;
    for (int k = li; k < ui; k++)
      if (env >= *sm++) present = 1;
    for (int k = li; k < ui; k++) {
      *ar++ = (int8_t)!present;
      if (k >= ui_noise) {
        nb++;
        ui_noise = h->freq_band_tbl_noise[nb + 1];
        nm = (int16_t)(noise_floor[nb] & 0xffc0);
        ne = (int16_t)((noise_floor[nb] & 63) - 38);
      }
      if (k >= f->max_qmf_subband_aac) {
        e_orig[2 * c] = ref_m;
        e_orig[2 * c + 1] = ref_e;
        sine[2 * c] = 0;
        sine[2 * c + 1] = 0;
        xs_subbandgain(ref_m, nm, est[2 * c], est[2 * c + 1], ne, ref_e, present, env >= sm1[c], noise_absc,
                       &gain[2 * c], &noise_lvl[2 * c], &sine[2 * c]);
        c++;
      }
    }
  }
}

/* env_calc.c:1454 */
FX_HD void xs_avggain(const int16_t *e_orig, const int16_t *est, int b0, int b1, int16_t *o_mant, int16_t *o_exp,
                      int16_t *avg_m, int16_t *avg_e, int flag) {
  int32_t som = 0, soe = 0, sem = 0, see = 0;
  for (int k = b0; k < b1; k++) {
    int16_t m = e_orig[2 * k], e = e_orig[2 * k + 1], m2 = est[2 * k], e2 = est[2 * k + 1];
    xs_acc_me(&som, &soe, m, e);
    if (flag) {
      m = (int16_t)(((int32_t)m * m2) >> 16);
      e = (int16_t)(e + e2 + 1);
    } else {
      m = m2;
      e = e2;
    }
    xs_acc_me(&sem, &see, m, e);
  }
  int nv = 16 - xs_pnorm32(som);
  if (nv > 0) {
    som >>= nv;
    soe += nv;
  }
  nv = 16 - xs_pnorm32(sem);
  if (nv > 0) {
    sem >>= nv;
    see += nv;
  }
  int16_t so_m, so_e, se_m, se_e;
  if (!flag) {
    so_m = (int16_t)som;
    so_e = (int16_t)soe;
    se_m = (int16_t)sem;
    se_e = (int16_t)see;
  } else {
    se_m = (int16_t)som;
    se_e = (int16_t)soe;
    so_m = (int16_t)sem;
    so_e = (int16_t)see;
  }
  int t = xs_fix_mant_div(so_m, se_m, avg_m);
  *avg_e = (int16_t)(t + (so_e - se_e) + 1);
  *o_mant = so_m;
  *o_exp = so_e;
}

/* env_calc.c:229 */
FX_HD void xs_noiselimiting(const xaac_sbr_header *h, int skip, const int16_t *e_orig, const int16_t *est,
                            int16_t *gain, int16_t *noise_lvl, int16_t *sine, const int16_t *lim_tab, int noise_absc) {
  const int16_t lim_m = lim_tab[0], lim_e = lim_tab[1];
  for (int c = 0; c < h->num_lf_bands; c++) {
    int b0 = 0, b1 = 0;
    if (h->freq_band_tbl_lim[c] > skip) b0 = h->freq_band_tbl_lim[c] - skip;
    if (h->freq_band_tbl_lim[c + 1] > skip) b1 = h->freq_band_tbl_lim[c + 1] - skip;
    if (b0 >= b1) continue;
    int16_t so_m, so_e, mg_m, mg_e;
    xs_avggain(e_orig, est, b0, b1, &so_m, &so_e, &mg_m, &mg_e, 0);
    int32_t mt = xs_mult16x16_shl(mg_m, lim_m);
    mg_e = (int16_t)(mg_e + lim_e);
    int tv = fx_norm32(mt);
    mg_e = (int16_t)(mg_e - tv);
    mg_m = (int16_t)(xs_shl(mt, tv) >> 16);
    if (mg_e >= 34) {
      mg_m = 0x3000;
      mg_e = 34;
    }
    for (int k = b0; k < b1; k++) {
      int16_t gm = gain[2 * k], ge = gain[2 * k + 1];
      if (ge > mg_e || (ge == mg_e && gm > mg_m)) {
        int16_t na_m;
        int na_e = xs_fix_mant_div(mg_m, gm, &na_m);
        na_e += (mg_e - ge) + 1;
        noise_lvl[2 * k] = (int16_t)(fx_shl_dir_sat_limit(xs_mult16x16_shl(noise_lvl[2 * k], na_m), (int16_t)na_e) >> 16);
        gain[2 * k] = mg_m;
        gain[2 * k + 1] = mg_e;
      }
    }
    int32_t am = 0, ae = 0;
    for (int k = b0; k < b1; k++) {
      int32_t m = ((int32_t)gain[2 * k] * est[2 * k]) >> 15;
      int32_t e = gain[2 * k + 1] + est[2 * k + 1];
      xs_acc_me(&am, &ae, m, e);
      if (sine[2 * k] != 0)
        xs_acc_me(&am, &ae, sine[2 * k], sine[2 * k + 1]);
      else if (noise_absc == 0)
        xs_acc_me(&am, &ae, noise_lvl[2 * k], noise_lvl[2 * k + 1]);
    }
    int nv = 16 - fx_norm32(am);
    if (nv > 0) {
      am >>= nv;
      ae += nv;
    }
    int16_t bg_m;
    int bg_e = xs_fix_mant_div(so_m, (int16_t)am, &bg_m);
    bg_e = (int16_t)(bg_e + (so_e - (int16_t)ae) + 1);
    if (bg_e > 2 || (bg_e == 2 && bg_m > 0x5061)) {
      bg_m = 0x5061;
      bg_e = 2;
    }
    for (int k = b0; k < b1; k++) {
      gain[2 * k] = xs_mult16_shl(gain[2 * k], bg_m);
      sine[2 * k] = xs_mult16_shl(sine[2 * k], bg_m);
      noise_lvl[2 * k] = xs_mult16_shl(noise_lvl[2 * k], bg_m);
      gain[2 * k + 1] = (int16_t)(gain[2 * k + 1] + bg_e);
      sine[2 * k + 1] = (int16_t)(sine[2 * k + 1] + bg_e);
      noise_lvl[2 * k + 1] = (int16_t)(noise_lvl[2 * k + 1] + bg_e);
    }
  }
}

/* env_calc.c:78 (low-power only) */
FX_HD void xs_alias_reduction(const int16_t *deg, int16_t *gain, const int16_t *est, const int8_t *alias_red, int nsb) {
  int16_t grp[XS_MAXF + 2];
  int grouping = 0, i = 0;
  for (int k = 0; k < nsb - 1; k++) {
    if (deg[k + 1] != 0 && alias_red[k]) {
      if (!grouping) {
        grp[i++] = (int16_t)k;
        grouping = 1;
      } else if (grp[i - 1] + 3 == k) {
        grp[i++] = (int16_t)(k + 1);
        grouping = 0;
      }
    } else if (grouping) {
      grouping = 0;
      grp[i] = (int16_t)k;
      if (alias_red[k]) grp[i] = (int16_t)(k + 1);
      i++;
    }
  }
  if (grouping) grp[i++] = (int16_t)nsb;
  const int ngroups = i >> 1;
  for (int g = 0; g < ngroups; g++) {
    const int b0 = grp[2 * g], b1 = grp[2 * g + 1];
    int16_t amp_m, amp_e, gg_m, gg_e;
    xs_avggain(est, gain, b0, b1, &amp_m, &amp_e, &gg_m, &gg_e, 1);
    int32_t mod_m = 0, mod_e = 0;
    for (int k = b0; k < b1; k++) {
      int16_t alpha = deg[k];
      if (k < nsb - 1 && deg[k + 1] > alpha) alpha = deg[k + 1];
      int32_t gain_m = (int32_t)alpha * gg_m;
      int16_t one_minus = (int16_t)(0x7fff - alpha);
      int32_t tm = gain[2 * k], te = gain[2 * k + 1];
      tm = ((int32_t)one_minus * tm) >> 15;
      int32_t d = gg_e - te;
      if (d >= 0) {
        te = gg_e;
        tm = fx_shr(tm, d);
        tm = (gain_m >> 15) + tm;
      } else {
        tm = fx_shr(gain_m, 15 - d) + tm;
      }
      gain[2 * k] = (int16_t)tm;
      gain[2 * k + 1] = (int16_t)te;
      /* the reference multiplies the untruncated 32-bit tmp_gain_mant here (env_calc.c:182) */
      int32_t pm = (int32_t)((uint32_t)tm * (uint32_t)(int32_t)est[2 * k]) >> 16;
      int32_t pe = te + est[2 * k + 1] + 1;
      xs_acc_me(&mod_m, &mod_e, pm, pe);
    }
    int nv = 16 - xs_pnorm32(mod_m);
    if (nv > 0) {
      mod_m >>= nv;
      mod_e += nv;
    }
    int16_t comp_m;
    int comp_e = xs_fix_mant_div(amp_m, (int16_t)mod_m, &comp_m);
    comp_e = (int16_t)(comp_e + amp_e - (int16_t)mod_e + 1 + 1);
    for (int k = b0; k < b1; k++) {
      gain[2 * k] = (int16_t)(((int32_t)gain[2 * k] * comp_m) >> 16);
      gain[2 * k + 1] = (int16_t)(gain[2 * k + 1] + comp_e);
    }
  }
}

/* env_calc.c:423 */
FX_HD void xs_erg_to_amplitude_lp(int bands, int16_t noise_e, int16_t *sine, int16_t *gain, int16_t *noise_lvl) {
  for (int k = 0; k < bands; k++) {
    xs_mant_exp_sqrt(&sine[2 * k]);
    xs_mant_exp_sqrt(&gain[2 * k]);
    xs_mant_exp_sqrt(&noise_lvl[2 * k]);
    int shift = (noise_e - noise_lvl[2 * k + 1]) - 4;
    if (shift > 0)
      noise_lvl[2 * k] = (int16_t)xs_sar(noise_lvl[2 * k], shift);
    else
      noise_lvl[2 * k] = (int16_t)xs_shl(noise_lvl[2 * k], -shift);
    shift = sine[2 * k + 1] - noise_e;
    if (shift > 0)
      sine[2 * k] = xs_shl16_sat(sine[2 * k], (int16_t)shift);
    else
      sine[2 * k] = (int16_t)xs_sar(sine[2 * k], (int16_t)-shift);
  }
}

/* env_calc.c:1017 */
FX_HD void xs_equalize_filt_buf(int16_t *fb, int16_t *gain, int n) {
  for (int b = 0; b < n; b++, fb += 2, gain += 2) {
    int32_t fe = fb[1], ge = gain[1], fm = fb[0], gm = gain[0];
    int32_t diff = ge - fe;
    if (diff >= 0) {
      fb[1] = (int16_t)ge;
      fb[0] = (int16_t)xs_sar(fb[0], diff);
    } else {
      int32_t reserve = fx_norm32(fm) - 16;
      if (diff + reserve >= 0) {
        fb[0] = (int16_t)xs_shl(fm, -diff);
        fb[1] = (int16_t)(fe + diff);
      } else {
        fb[0] = (int16_t)xs_shl(fm, reserve);
        fb[1] = (int16_t)(fe - reserve);
        int32_t shift = -(reserve + diff);
        gain[0] = (int16_t)xs_sar(gm, shift);
        gain[1] = (int16_t)(gain[1] + shift);
      }
    }
  }
}

/* env_calc.c:1080 */
FX_HD void xs_noise_rescale(int16_t *p, int diff, int n, int step) {
  if (diff > 0)
    for (int k = 0; k < n; k++) p[k * step] = (int16_t)xs_sar(p[k * step], diff);
  else if (diff < 0)
    for (int k = 0; k < n; k++) p[k * step] = (int16_t)xs_shl(p[k * step], -diff);
}

#define XS_FACTOR ((int32_t)(0x010b0000 * 2))

/* env_calc.c:1564: one slot, harmonic index 0 / 2 */
FX_HD void xs_harm_zerotwo_lp(const XsQmf &x, int slot, int b0, const int16_t *gain, int scale_change,
                              const int16_t *sine, const int32_t *rand_ph, const int16_t *noise_lvl, int nsb,
                              int noise_absc, int harm_index) {
  scale_change -= 1;
  for (int k = 0; k < nsb; k++) {
    int32_t v = fx_mul32x16(x(slot, b0 + k), gain[2 * k]);
    int shift = gain[2 * k + 1] - scale_change;
    v = shift > 0 ? xs_shl(v, shift) : xs_sar(v, -shift);
    int32_t sl = xs_shl(sine[2 * k], 16);
    if (!noise_absc && sl == 0)
      v = xs_mac16x16_shl_sat(v, (int16_t)(rand_ph[k] >> 16), noise_lvl[2 * k]);
    else if (harm_index == 0)
      v = fx_add_sat(v, sl);
    else
      v = fx_sub_sat(v, sl);
    x(slot, b0 + k) = v;
  }
}

/* env_calc.c:1617: one slot, harmonic index 1 / 3 */
FX_HD void xs_harm_onethree_lp(const XsQmf &x, int slot, int b0, const int16_t *gain, int scale_change,
                               const int16_t *sine, const int32_t *rand_ph, const int16_t *noise_lvl, int nsb,
                               int noise_absc, int freq_inv, int noise_e, int sb_start) {
  int k = 0, tone_count = 0;
  scale_change -= 1;
  int32_t v = fx_mul32x16(x(slot, b0), gain[0]);
  int shift = gain[1] - scale_change;
  v = shift > 0 ? xs_shl(v, shift) : xs_sar(v, -shift);
  int16_t sl = sine[0], sl_prev, sl_next = nsb > 1 ? sine[2] : (int16_t)0;
  if (sine[0] != 0)
    tone_count++;
  else if (!noise_absc)
    v = xs_mac16x16_shl_sat(v, (int16_t)(rand_ph[0] >> 16), noise_lvl[0]);
  int32_t tm2 = fx_mul32x16(XS_FACTOR, sl_next);
  int32_t tm = fx_mul32x16(XS_FACTOR, sl);
  int16_t ne = (int16_t)noise_e;
  tm = ne > 0 ? fx_shl(tm, ne) : fx_shr(tm, -ne);
  if (freq_inv < 0) {
    x(slot, b0 - 1) = fx_add_sat(x(slot, b0 - 1), tm);
    v = fx_sub_sat(v, tm2);
  } else {
    x(slot, b0 - 1) = fx_sub_sat(x(slot, b0 - 1), tm);
    v = fx_add_sat(v, tm2);
  }
  x(slot, b0) = v;
  const int nm1 = nsb - 1;
  for (k = 1; k < nm1; k++) {
    v = fx_mul32x16(x(slot, b0 + k), gain[2 * k]);
    shift = gain[2 * k + 1] - scale_change;
    v = shift >= 0 ? xs_shl(v, shift) : xs_sar(v, -shift);
    sl_prev = sl;
    sl = sl_next;
    if (sl != 0) tone_count++;
    sl_next = sine[2 * (k + 1)];
    if (!noise_absc && sl == 0) v = xs_mac16x16_shl_sat(v, (int16_t)(rand_ph[k] >> 16), noise_lvl[2 * k]);
    if (tone_count <= 16) {
      int32_t add = fx_mul32x16(XS_FACTOR, (int16_t)(sl_prev - sl_next));
      v = fx_add_sat(v, (int32_t)((uint32_t)add * (uint32_t)freq_inv));
    }
    x(slot, b0 + k) = v;
    freq_inv = -freq_inv;
  }
  freq_inv = (freq_inv + 1) >> 1;
  if (nm1 > 0) {
    v = fx_mul32x16(x(slot, b0 + k), gain[2 * k]);
    shift = gain[2 * k + 1] - scale_change;
    v = shift > 0 ? xs_shl(v, shift) : xs_sar(v, -shift);
    int32_t tms = fx_mul32x16(XS_FACTOR, sl);
    sl = sl_next;
    if (sl != 0)
      tone_count++;
    else if (!noise_absc)
      v = xs_mac16x16_shl_sat(v, (int16_t)(rand_ph[k] >> 16), noise_lvl[2 * k]);
    if (tone_count <= 16) {
      tm2 = fx_mul32x16(XS_FACTOR, sl);
      if (freq_inv) {
        x(slot, b0 + k) = fx_add_sat(v, tms);
        if (k + sb_start < 62) x(slot, b0 + k + 1) = fx_sub_sat(x(slot, b0 + k + 1), tm2);
      } else {
        x(slot, b0 + k) = fx_sub_sat(v, tms);
        if (k + sb_start < 62) x(slot, b0 + k + 1) = fx_add_sat(x(slot, b0 + k + 1), tm2);
      }
    } else {
      x(slot, b0 + k) = v;
    }
  }
}

/* env_calc.c:479, low-power branch: apply gains / noise / sines to slots [s0,s1) */
FX_HD void xs_adapt_noise_gain_lp(xaac_sbr_state *st, int noise_e, int nsb, int skip, int16_t *gain, int16_t *noise_lvl,
                                  int16_t *sine, int s0, int s1, int input_e, int adj_e, int final_e, int sb_start,
                                  int lb_scale, int noise_absc, const XsQmf &x) {
  const int bands = nsb - skip;
  if (st->start_up) {
    st->start_up = 0;
    st->filt_buf_noise_e = noise_e;
    for (int k = 0; k < bands; k++) {
      st->filt_buf_me[2 * (skip + k)] = gain[2 * k];
      st->filt_buf_me[2 * (skip + k) + 1] = gain[2 * k + 1];
      st->filt_buf_noise_m[skip + k] = noise_lvl[2 * k];
    }
  } else {
    xs_equalize_filt_buf(&st->filt_buf_me[2 * skip], gain, bands);
  }
  for (int l = s0; l < s1; l++) {
    int scale_change;
    if (l < 32) {
      scale_change = adj_e - input_e;
    } else {
      scale_change = final_e - input_e;
      if (l == 32 && s0 < 32) {
        int diff = final_e - noise_e;
        noise_e = final_e;
        xs_noise_rescale(noise_lvl, diff, bands, 2);
      }
    }
    xs_noise_rescale(st->filt_buf_noise_m, st->filt_buf_noise_e - noise_e, nsb, 1);
    st->filt_buf_noise_e = noise_e;
    const int index = st->ph_index, harm_index = st->harm_index;
    const int32_t *rp = &xaac_sbr_rand_ph[index + 1];
    st->ph_index = (int16_t)((index + nsb) & 511);
    st->harm_index = (int16_t)((harm_index + 1) & 3);
    if (!(harm_index & 1)) {
      xs_harm_zerotwo_lp(x, l, sb_start, gain, scale_change, sine, rp, noise_lvl, nsb, noise_absc, harm_index);
    } else {
      int noise = (noise_e - 16) - lb_scale;
      int fi = !(sb_start & 1);
      fi = (fi << 1) - 1;
      if (harm_index == 3) fi = -fi;
      xs_harm_onethree_lp(x, l, sb_start, gain, scale_change, sine, rp, noise_lvl, nsb, noise_absc, fi, noise,
                          sb_start);
    }
  }
  for (int k = 0; k < bands; k++) {
    st->filt_buf_me[2 * (skip + k)] = gain[2 * k];
    st->filt_buf_noise_m[skip + k] = noise_lvl[2 * k];
  }
}

/* env_calc.c:692, low-power, AAC-LC/HE-AAC (not ELD), 1024-sample frames.  Returns 0 or -1. */
FX_HD int xs_calc_sbrenvelope_lp(const xaac_sbr_header *h, const xaac_sbr_frame *f, xaac_sbr_state *st,
                                 const XsQmf &x, const int16_t *deg_patched) {
  const int num_env = f->num_env;
  const int16_t *border = f->border_vec;
  const int16_t *noise_floor = f->int_noise_floor;
  const int sb_start = h->sub_band_start, sb_end = h->sub_band_end;
  const int nsb = sb_end - sb_start;
  const int skip = f->max_qmf_subband_aac - sb_start;
  int16_t nrg_est[2 * XS_MAXF], nrg_gain[2 * XS_MAXF], noise_lvl[2 * XS_MAXF], nrg_sine[2 * XS_MAXF],
      e_orig[2 * XS_MAXF];
  int8_t sine_mapped[XS_MAXF], alias_red[64];
  for (int i = 0; i < 2 * XS_MAXF; i++) nrg_est[i] = nrg_gain[i] = noise_lvl[i] = nrg_sine[i] = e_orig[i] = 0;
  for (int i = 0; i < 64; i++) alias_red[i] = 0;
  xs_map_sineflags(h->freq_band_tbl_hi, h->num_sf_bands[1], f->add_harmonics, st->harm_flags_prev, f->transient_env,
                   sine_mapped);
  int adj_e;
  {
    int first_band = (st->prev_max_qmf_subband_aac > f->max_qmf_subband_aac ? st->prev_max_qmf_subband_aac
                                                                             : f->max_qmf_subband_aac) - sb_start;
    int16_t max_noise = 0;
    for (int i = first_band; i < nsb; i++)
      if (st->filt_buf_noise_m[i] > max_noise) max_noise = st->filt_buf_noise_m[i];
    adj_e = (st->filt_buf_noise_e - fx_norm32(max_noise)) - 16;
  }
  int final_e = 0;
  {
    const int16_t *p = f->int_env_sf_arr;
    for (int i = 0; i < num_env; i++) {
      int mx = 16 - 16; /* NRG_EXP_OFFSET - SHORT_BITS */
      const int fr = f->freq_res[i];
      for (int j = 0; j < h->num_sf_bands[fr]; j++) {
        int t = *p++ & 63;
        if (t > mx) mx = t;
      }
      mx -= 16;
      int t = (mx + 13) >> 1;
      if (border[i] < 16 && t > adj_e) adj_e = (int16_t)t;
      if (border[i + 1] > 16 && t > final_e) final_e = (int16_t)t;
    }
  }
  int m = 0, nf_idx = 0;
  for (int i = 0; i < num_env; i++) {
    const int s0 = 2 * border[i], s1 = 2 * border[i + 1];
    if (s0 >= 38 || s1 > 38) return -1;
    const int fr = f->freq_res[i];
    if (nf_idx >= XAAC_SBR_MAX_NOISE_ENVELOPES) return -1;
    if (border[i] == f->noise_border_vec[nf_idx + 1]) {
      noise_floor += h->num_nf_bands;
      nf_idx++;
    }
    int noise_absc;
    if (i == f->transient_env || i == st->tansient_env_prev)
      noise_absc = 1;
    else
      noise_absc = 0;
    const int input_e = 15 - st->hb_scale;
    if (h->interpol_freq)
      xs_energy_per_subband(x, s0, s1, f->max_qmf_subband_aac, sb_end, input_e, nrg_est);
    else
      xs_energy_per_sfb(x, h->num_sf_bands[fr], fr ? h->freq_band_tbl_hi : h->freq_band_tbl_lo, s0, s1,
                        f->max_qmf_subband_aac, input_e, nrg_est);
    if ((fr ? h->freq_band_tbl_hi : h->freq_band_tbl_lo)[0] < sb_start) return -1;
    xs_calc_subband_gains(h, f, fr, noise_floor, h->num_sf_bands[fr], m, i, sine_mapped, alias_red, e_orig, nrg_sine,
                          nrg_est, nrg_gain, noise_lvl, noise_absc);
    m += h->num_sf_bands[fr];
    xs_noiselimiting(h, skip, e_orig, nrg_est, nrg_gain, noise_lvl, nrg_sine, &xaac_sbr_lim_gains_m[2 * h->limiter_gains],
                     noise_absc);
    xs_alias_reduction(deg_patched + sb_start, nrg_gain, nrg_est, alias_red, nsb);
    const int16_t noise_e = (int16_t)(s0 < 32 ? adj_e : final_e);
    const int bands = nsb - skip;
    xs_erg_to_amplitude_lp(bands, noise_e, nrg_sine, nrg_gain, noise_lvl);
    const int16_t lb_scale = (int16_t)(15 - st->lb_scale);
    xs_adapt_noise_gain_lp(st, noise_e, nsb, skip, nrg_gain, noise_lvl, nrg_sine, s0, s1, input_e, adj_e, final_e,
                           f->max_qmf_subband_aac, lb_scale, noise_absc, x);
  }
  const int first_start = border[0] * 2;
  {
    const int ov_adj_e = 15 - st->ov_hb_scale;
    const int output_e = ov_adj_e > adj_e ? ov_adj_e : adj_e; /* reserves are 0 without PS */
    xs_adjust(x, f->max_qmf_subband_aac, sb_end, 0, first_start, ov_adj_e - output_e);
    xs_adjust(x, f->max_qmf_subband_aac, sb_end, first_start, h->num_time_slots * h->time_step, adj_e - output_e);
    st->hb_scale = (int16_t)(15 - output_e);
  }
  st->ov_hb_scale = (int16_t)(15 - final_e);
  st->tansient_env_prev = (f->transient_env == num_env) ? 0 : -1;
  return 0;
}

/* sbrdec_lpfuncs.c:453 (real-valued) */
FX_HD void xs_rescale_x_overlap(const xaac_sbr_header *h, const xaac_sbr_frame *f, xaac_sbr_state *st, const XsQmf &x) {
  const int old_lsb = st->prev_max_qmf_subband_aac;
  const int start_slot = h->time_step * (st->prev_end_position - h->num_time_slots);
  const int new_lsb = f->max_qmf_subband_aac;
  st->codec_usb = (int16_t)new_lsb;
  st->syn_lsb = (int16_t)new_lsb;
  int b0 = old_lsb < new_lsb ? old_lsb : new_lsb, b1 = old_lsb < new_lsb ? new_lsb : old_lsb;
  if (new_lsb == old_lsb || old_lsb <= 0) return;
  for (int l = start_slot; l < 6; l++)
    for (int k = old_lsb; k < new_lsb; k++) x(l, k) = 0;
  int source, target, t_lsb, t_usb;
  if (new_lsb > old_lsb) {
    source = st->ov_hb_scale;
    target = st->ov_lb_scale;
    t_lsb = 0;
    t_usb = old_lsb;
  } else {
    source = st->ov_lb_scale;
    target = st->ov_hb_scale;
    t_lsb = old_lsb;
    t_usb = st->syn_usb;
  }
  const int reserve = xs_headroom(x, b0, b1, 0, start_slot);
  xs_adjust(x, b0, b1, 0, start_slot, reserve);
  source += reserve;
  int delta = target - source;
  if (delta > 0) {
    delta = -delta;
    b0 = t_lsb;
    b1 = t_usb;
    if (new_lsb > old_lsb)
      st->ov_lb_scale = (int16_t)source;
    else
      st->ov_hb_scale = (int16_t)source;
  }
  xs_adjust(x, b0, b1, 0, start_slot, delta);
}

/* The part of ixheaacd_sbr_dec between the two QMF banks (sbr_dec.c:1050-1245, low-power mode).
   On entry x holds the 6 overlap slots (already through xs_rescale_x_overlap) and the 32 freshly
   analysed slots (bands 0..31); on exit x is ready for the synthesis bank and the state carries the
   new scale factors, LPC history and envelope-adjuster memory.  Returns 0 or -1. */
FX_HD int xs_sbr_core_lp(const xaac_sbr_header *h, const xaac_sbr_frame *f, xaac_sbr_state *st, const XsQmf &x,
                         int *save_lb_scale_out) {
  const int usb = st->codec_usb;
  XsMat<int32_t> lpc = {&st->lpc_real[0][0], 1}; /* two rows of 32 */
  int reserve = xs_headroom(x, 0, usb, 6, 38);
  int reserve_ov1 = xs_headroom(x, 0, usb, 0, 6);
  const int max_samp_val = reserve < reserve_ov1 ? reserve : reserve_ov1;
  int32_t m = 1;
  for (int i = 0; i < 2; i++)
    for (int k = 0; k < usb; k++) m |= fx_abs_nrm(st->lpc_real[i][k]);
  const int reserve_ov2 = xs_pnorm32(m);
  if (reserve_ov2 < reserve_ov1) reserve_ov1 = reserve_ov2;
  const int shift1 = st->lb_scale + reserve, shift2 = st->ov_lb_scale + reserve_ov1;
  const int min_shift = shift1 < shift2 ? shift1 : shift2;
  const int shift_over = shift2 - min_shift;
  reserve -= (shift1 - min_shift);
  st->ov_lb_scale = (int16_t)(st->ov_lb_scale + (reserve_ov1 - shift_over));
  xs_adjust(x, 0, usb, 0, 6, reserve_ov1 - shift_over);
  xs_adjust(x, 0, usb, 6, 38, reserve);
  {
    int sh = reserve_ov1 - shift_over;
    if (sh != 0) {
      if (sh > 31) sh = 31;
      if (sh < -31) sh = -31;
      for (int i = 0; i < 2; i++)
        for (int k = 0; k < usb; k++)
          st->lpc_real[i][k] = sh > 0 ? fx_shlw(st->lpc_real[i][k], sh) : (st->lpc_real[i][k] >> -sh);
    }
  }
  (void)lpc;
  st->lb_scale = (int16_t)(st->lb_scale + reserve);
  const int save_lb_scale = st->lb_scale;
  *save_lb_scale_out = save_lb_scale;
  for (int l = 6; l < 38; l++)
    for (int k = 32; k < 64; k++) x(l, k) = 0;
  if (f->apply_processing) {
    int16_t degree_alias[64];
    for (int k = 0; k < 64; k++) degree_alias[k] = 0;
    const int16_t last = fx_sat16((int32_t)f->border_vec[f->num_env] - h->num_time_slots);
    xs_low_pow_hf_generator(h, st, x, degree_alias, f->border_vec[0] * h->time_step, h->time_step * last,
                            f->max_qmf_subband_aac, f->sbr_invf_mode, st->prev_invf_mode, max_samp_val);
    st->hb_scale = (int16_t)((st->ov_lb_scale < st->lb_scale ? st->ov_lb_scale : st->lb_scale) - 2);
    if (xs_calc_sbrenvelope_lp(h, f, st, x, degree_alias)) return -1;
    for (int i = 0; i < h->num_if_bands; i++) st->prev_invf_mode[i] = f->sbr_invf_mode[i];
    st->prev_coupling_mode = f->coupling_mode;
    st->prev_max_qmf_subband_aac = f->max_qmf_subband_aac;
    st->prev_end_position = f->border_vec[f->num_env];
    st->prev_amp_res = f->amp_res;
  } else {
    st->hb_scale = (int16_t)save_lb_scale;
  }
  for (int i = 0; i < 2; i++)
    for (int k = 0; k < st->codec_usb; k++) st->lpc_real[i][k] = x(30 + i, k);
  return 0;
}

#endif /* XAAC_SBR_CORE_H */
