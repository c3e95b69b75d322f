/*
 * esbr_core.h -- the float HF generator and envelope adjuster of the reference's default SBR path ("Path A", -esbr:1)
 * for HE-AAC streams, shared by the gfx950 kernel (esbr_core_kernel.hip: one wave = one channel-frame, lane = QMF band)
 * and, compiled for the host with one "lane", by the checker (oracle/oracle_esbr.cpp).
 *
 * Restates
 *   ixheaacd_generate_hf            decoder/ixheaacd_sbrdec_lpfuncs.c:981   (LPP transposer: covariance :781, chirp :832)
 *   ixheaacd_sbr_env_calc           decoder/ixheaacd_esbr_envcal.c:71       (ORIG_SBR branch :640-860)
 *   ixheaacd_createlimiterbands     esbr_envcal.c:910,  ixheaacd_apply_inter_tes  esbr_envcal.c:1021
 *   ixheaacd_esbr_synthesis_regrp   decoder/ixheaacd_sbr_dec.c:297 (stereo_config_idx <= 0)
 * for usac_flag = 0: 2:1 SBR, no harmonic transposer, no PVC, no pre-flattening, no MPS.
 *
 * Float results depend on the order of every operation, so every expression keeps the reference's operand order and
 * width (float unless the reference mixes in a double: the 1e-17 guard, the sqrt calls); sums over bands or slots run
 * sequentially in the reference's order on whichever lane needs them.  Compiled with -ffp-contract=off on both sides.
 */
#ifndef XAAC_ESBR_CORE_H
#define XAAC_ESBR_CORE_H

#include <math.h>

#include "../../include/xaac_esbr.h"
#include "sbr_core.h" /* XsCx, XS_PAR, XS_ONE */
#include "fx_libm.h"

#if defined(__HIPCC__)
#define XAAC_TAB_QUAL static __device__ const
#include "tables_esbr.inc"
#undef XAAC_TAB_QUAL
#else
#include "tables_esbr.inc"
#endif

#pragma clang fp contract(off)

#ifndef XE_T
#define XE_T(i) /* optional stage timer hook (tools/prof_esbr_core.py) */
#endif
#ifndef XE_RANDOM_PHASE
#define XE_RANDOM_PHASE(i) xaac_esbr_random_phase[i] /* the kernel reads its LDS copy */
#endif

/* a [rows][64] float matrix pair; row 0 is the reference's pointer + SBR_HF_ADJ_OFFSET */
struct XeMat {
  float *re, *im;
  FX_MEMBER float &r(int l, int k) const { return re[l * 64 + k]; }
  FX_MEMBER float &i(int l, int k) const { return im[l * 64 + k]; }
};

/* Column walks fetch / write back XE_CH rows in one burst, so that a stage costs one memory latency per XE_CH rows
   instead of one per row (the compiler may not move a load across an earlier store of the same matrix by itself).  The
   order of the arithmetic is untouched. */
#ifndef XE_CH
#define XE_CH 8
#endif
#if defined(__HIPCC__)
#define XE_UNROLL _Pragma("unroll")
#define XE_NOUNROLL _Pragma("nounroll")
#else
#define XE_UNROLL
#define XE_NOUNROLL
#endif
FX_HD void xe_rows_load(const XeMat &m, int k, int r0, int r1, float *cr, float *ci) { /* rows r0 .. min(r0 + XE_CH, r1) - 1 */
  XE_UNROLL
  for (int j = 0; j < XE_CH; j++)
    if (r0 + j < r1) {
      cr[j] = m.r(r0 + j, k);
      ci[j] = m.i(r0 + j, k);
    }
}
FX_HD void xe_rows_store(const XeMat &m, int k, int r0, int r1, const float *cr, const float *ci) {
  XE_UNROLL
  for (int j = 0; j < XE_CH; j++)
    if (r0 + j < r1) {
      m.r(r0 + j, k) = cr[j];
      m.i(r0 + j, k) = ci[j];
    }
}

struct XeWork {
  float alpha_r[64][2], alpha_i[64][2];
  float bw_array[XAAC_SBR_MAX_PATCHES];
  float nrg_est[64], nrg_ref[64], nrg_gain[64], noise_level[64], nrg_tone[64];
  float pow_lo[80], pow_hi[80], tes_gain[80];
  float lim_pref[13], lim_gmax[13];
  int32_t err;
  int16_t src_band[64]; /* HF generator: source band of high band k2; -1: cleared; -2: not in a patch */
  int8_t bw_idx[64];
  int8_t harmonics[64];
  int8_t sfb_first[64], sfb_len[64], flag[64], o_idx[64], lim_of[64];
  int16_t m_idx[64];
  /* copies of state members the envelope loop reads band by band (from global memory each read was a memory latency) */
  int32_t lim_tab[13];
  int8_t harm_prev[64];
  int8_t own_tone[64]; /* per envelope: band c carries a sinusoid that counts in this envelope */
};

struct XeTrue { static constexpr bool value = true; };
struct XeFalse { static constexpr bool value = false; };

FX_HD double xe_sqrt(double v) { return sqrt(v); }

/* ---- ixheaacd_createlimiterbands (b_patching_mode = 1): serial, integer, run at a reset frame ------------------- */
FX_HD void xe_shellsort(int32_t *in, int n) { /* esbr_envcal.c:48: any sort gives the same array of integers */
  for (int i = 1; i < n; i++) {
    const int32_t v = in[i];
    int j = i;
    for (; j > 0 && in[j - 1] > v; j--) in[j] = in[j - 1];
    in[j] = v;
  }
}
/* x_over_qmf: the harmonic transposer's cross-over bands, or NULL; with harmonic patching they are the patch borders
   (esbr_envcal.c:929-941) */
/* tmp: 48 words of work space (the sorted band list and the patch borders: indexed at run time, so in the caller's
   shared memory, not in a lane's registers) */
FX_HD int xe_limiter_bands(const xaac_sbr_header *h, xaac_esbr_state *st, int harmonic, const int32_t *x_over_qmf, int32_t *tmp) {
  const int nb = h->num_sf_bands[0];
  const int16_t *tbl = h->freq_band_tbl_lo;
  const int sb_start = tbl[0], sb_end = tbl[nb];
  int num_patches = st->num_patches;
  int32_t *patch_borders = tmp /* [XAAC_SBR_MAX_PATCHES + 2] */, *t = tmp + XAAC_SBR_MAX_PATCHES + 2 /* [32 + XAAC_SBR_MAX_PATCHES + 1] */;
  int i;
  if (harmonic && x_over_qmf) {
    num_patches = 0;
    for (i = 1; i < 4; i++)
      if (x_over_qmf[i] != 0) num_patches++;
    for (i = 0; i < num_patches; i++) patch_borders[i] = x_over_qmf[i] - sb_start;
  } else {
    if (num_patches < 0 || num_patches > XAAC_SBR_MAX_PATCHES) return -1;
    for (i = 0; i < num_patches; i++) patch_borders[i] = st->patch_start_subband[i] - sb_start;
  }
  patch_borders[i] = sb_end - sb_start;
  st->lim_table[0][0] = tbl[0] - sb_start;
  st->lim_table[0][1] = tbl[nb] - sb_start;
  st->gate_mode[0] = 1;
  for (i = 1; i < 4; i++) {
    for (int k = 0; k <= nb; k++) t[k] = tbl[k] - sb_start;
    for (int k = 1; k < num_patches; k++) t[nb + k] = patch_borders[k];
    int gate = nb + num_patches - 1;
    xe_shellsort(t, gate + 1);
    for (int j = 1; j <= gate; j++) {
      const int a = t[j] + sb_start, b = t[j - 1] + sb_start;
      const bool close = a >= 1 && a <= 64 && b >= 1 && b <= 64 && ((xaac_esbr_lim_close[64 * (i - 1) + a - 1] >> (b - 1)) & 1);
      if (!close) continue;
      if (t[j] == t[j - 1]) {
        t[j] = sb_end;
        xe_shellsort(t, gate + 1);
        gate--;
        j--;
        continue;
      }
      bool pb0 = false, pb1 = false;
      for (int k = 0; k <= num_patches; k++) pb0 = pb0 || t[j - 1] == patch_borders[k];
      for (int k = 0; k <= num_patches; k++) pb1 = pb1 || t[j] == patch_borders[k];
      if (!pb1) {
        t[j] = sb_end;
        xe_shellsort(t, gate + 1);
        gate--;
        j--;
      } else if (!pb0) {
        t[j - 1] = sb_end;
        xe_shellsort(t, gate + 1);
        gate--;
        j--;
      }
    }
    if (gate > 12) return -1;
    st->gate_mode[i] = gate;
    for (int k = 0; k <= gate; k++) st->lim_table[i][k] = t[k];
  }
  return 0;
}

/* Side info that would index past the boundary structs or the 40-row history (no legal HE-AAC frame does): refused.
   One lane's worth of integer checks; every lane computes the same answer. */
FX_HD int xe_side_info_bad(const xaac_sbr_header *h, const xaac_sbr_frame *f, const xaac_esbr_side *sd) {
  int bad = 0;
  bad |= f->num_env < 1 || f->num_env > XAAC_SBR_MAX_ENVELOPES;
  bad |= f->num_noise_env < 1 || f->num_noise_env > XAAC_SBR_MAX_NOISE_ENVELOPES;
  bad |= h->num_sf_bands[0] < 1 || h->num_sf_bands[0] > XAAC_SBR_MAX_FREQ_COEFFS / 2;
  bad |= h->num_sf_bands[1] < 1 || h->num_sf_bands[1] > XAAC_SBR_MAX_FREQ_COEFFS;
  bad |= h->num_nf_bands < 1 || h->num_nf_bands > XAAC_SBR_MAX_NOISE_COEFFS;
  bad |= sd->num_mf_bands < 1 || sd->num_mf_bands > XAAC_SBR_MAX_FREQ_COEFFS;
  bad |= h->sub_band_start < 1 || h->sub_band_start > 32 || h->sub_band_end < h->sub_band_start || h->sub_band_end > 64;
  bad |= sd->qmf_sb_prev < 0 || sd->qmf_sb_prev > 64 || sd->out_sampling_freq < 8000; /* 64: no SBR frame before this one */
  if (bad) return 1;
  for (int i = 0; i <= f->num_env; i++) bad |= f->border_vec[i] < 0 || f->border_vec[i] > 19;
  bad |= f->border_vec[f->num_env] < 16; /* rows the synthesis regrouping needs (the reference always has them) */
  for (int i = 0; i <= f->num_noise_env; i++) bad |= f->noise_border_vec[i] < 0 || f->noise_border_vec[i] > 19;
  for (int i = 0; i <= h->num_sf_bands[0]; i++) bad |= h->freq_band_tbl_lo[i] < 1 || h->freq_band_tbl_lo[i] > 64;
  for (int i = 0; i <= h->num_sf_bands[1]; i++) bad |= h->freq_band_tbl_hi[i] < 1 || h->freq_band_tbl_hi[i] > 64;
  for (int i = 0; i < h->num_sf_bands[0]; i++) bad |= h->freq_band_tbl_lo[i] > h->freq_band_tbl_lo[i + 1];
  for (int i = 0; i < h->num_sf_bands[1]; i++) bad |= h->freq_band_tbl_hi[i] > h->freq_band_tbl_hi[i + 1];
  for (int i = 0; i <= h->num_nf_bands; i++) bad |= h->freq_band_tbl_noise[i] < 1 || h->freq_band_tbl_noise[i] > 64;
  for (int i = 0; i < h->num_nf_bands; i++) bad |= h->freq_band_tbl_noise[i] >= h->freq_band_tbl_noise[i + 1];
  bad |= h->freq_band_tbl_noise[0] != h->sub_band_start || h->freq_band_tbl_noise[h->num_nf_bands] != h->sub_band_end;
  for (int i = 0; i <= sd->num_mf_bands; i++) bad |= sd->f_master_tbl[i] < 1 || sd->f_master_tbl[i] > 64;
  for (int i = 0; i < sd->num_mf_bands; i++) bad |= sd->f_master_tbl[i] > sd->f_master_tbl[i + 1];
  bad |= sd->f_master_tbl[0] > h->sub_band_start || sd->f_master_tbl[0] > 32;
  bad |= h->freq_band_tbl_lo[0] != h->sub_band_start || h->freq_band_tbl_hi[0] != h->sub_band_start;
  bad |= h->freq_band_tbl_lo[h->num_sf_bands[0]] != h->sub_band_end || h->freq_band_tbl_hi[h->num_sf_bands[1]] != h->sub_band_end;
  return bad;
}

/* ---- ixheaacd_generate_hf ---------------------------------------------------------------------------------------- */
FX_HD int xe_closest_entry(int goal, const int16_t *f, int n) { /* sbrdec_lpfuncs.c:263, direction 0 */
  if (goal <= f[0]) return f[0];
  if (goal >= f[n]) return f[n];
  int idx = n;
  while (f[idx] > goal) idx--;
  return f[idx];
}

/* the patch map (header-level integers, sbrdec_lpfuncs.c:1122-1200): lane 0 */
FX_HD void xe_build_patches(const xaac_sbr_header *h, const xaac_esbr_side *sd, xaac_esbr_state *st, XeWork *w) {
  const int16_t *fm = sd->f_master_tbl;
  const int nmf = sd->num_mf_bands;
  const int lsb = fm[0], usb = fm[nmf], xover_offset = h->sub_band_start - fm[0];
  for (int k = 0; k < 64; k++) w->src_band[k] = -2;
  int goal_sb = (int)(2.048e6f / (float)sd->out_sampling_freq + 0.5f);
  if (goal_sb < fm[nmf]) {
    int index = 0;
    while (fm[index] < goal_sb) index++;
    goal_sb = fm[index];
  } else {
    goal_sb = fm[nmf];
  }
  int source_start_band = xover_offset + 1, sb = lsb + xover_offset, patch = 0, flag_break = 0;
  while (sb < usb) {
    if (patch >= XAAC_SBR_MAX_PATCHES) {
      w->err = -1;
      return;
    }
    st->patch_start_subband[patch] = sb;
    int num = goal_sb - sb, stride;
    if (num >= lsb - source_start_band) {
      stride = (sb - source_start_band) & ~1;
      num = lsb - (sb - stride);
      num = xe_closest_entry(sb + num, fm, nmf) - sb;
    }
    stride = (num + sb - lsb + 1) & ~1;
    source_start_band = 1;
    if (goal_sb - (sb + num) < 3) goal_sb = usb;
    if (num < 3 && patch > 0 && sb + num == usb) {
      for (int k2 = sb; k2 < sb + num; k2++) w->src_band[k2] = -1;
      break;
    }
    if (num < 0 && flag_break == 1) break;
    if (num < 0) {
      flag_break = 1;
      continue;
    }
    flag_break = 0;
    if (sb - stride < 0 || sb + num > 64) { /* a source band in front of the matrix: the reference would read there */
      w->err = -1;
      return;
    }
    for (int k2 = sb; k2 < sb + num; k2++) w->src_band[k2] = (int16_t)(k2 - stride);
    sb += num;
    patch++;
  }
  st->num_patches = patch;
}

/* the inverse-filtering band of a patched band (sbrdec_lpfuncs.c:1201-1218), every lane for its own band */
FX_HD void xe_patch_bw_index(const XsCx &cx, const xaac_sbr_header *h, XeWork *w) {
  const int16_t *invf_tbl = h->freq_band_tbl_noise + 1;
  XS_PAR(k2, 0, 64) {
    if (w->src_band[k2] >= 0) {
      int bw_index = 0;
      while (bw_index < XAAC_SBR_MAX_NOISE_COEFFS && k2 >= invf_tbl[bw_index]) bw_index++;
      if (bw_index >= XAAC_SBR_MAX_NOISE_COEFFS) w->err = -1;
      else w->bw_idx[k2] = (int8_t)bw_index;
    }
  }
}

/* covariance of band k over 38 slots from row -2 on (ixheaacd_esbr_calc_co_variance, sbrdec_lpfuncs.c:781) and the
   second-order prediction coefficients of :1077-1120 */
FX_HD void xe_covar_alpha(const XeMat &src, int k, float &a0r, float &a0i, float &a1r, float &a1i, int len = 38 /* co_var_len: 76 for 4:1 SBR, :1040 */) {
  a0r = a0i = a1r = a1i = 0;
  float p01r = 0, p01i = 0, p02r = 0, p02i = 0, p11 = 0, p12r = 0, p12i = 0, p22 = 0;
  float r2 = src.r(-2, k), i2 = src.i(-2, k), r1 = src.r(-1, k), i1 = src.i(-1, k);
  XE_NOUNROLL
  for (int j0 = 0; j0 < len; j0 += XE_CH) {
    float cr[XE_CH] = {0}, ci[XE_CH] = {0};
    xe_rows_load(src, k, j0, len, cr, ci);
    XE_UNROLL
    for (int jj = 0; jj < XE_CH; jj++) if (j0 + jj < len) {
    const float r0 = cr[jj], i0 = ci[jj];
    p01r += r0 * r1 + i0 * i1;
    p01i += i0 * r1 - r0 * i1;
    p02r += r0 * r2 + i0 * i2;
    p02i += i0 * r2 - r0 * i2;
    p11 += r1 * r1 + i1 * i1;
    p12r += r1 * r2 + i1 * i2;
    p12i += i1 * r2 - r1 * i2;
    p22 += r2 * r2 + i2 * i2;
    r2 = r1; i2 = i1;
    r1 = r0; i1 = i0;
    }
  }
  const float det = p11 * p22 - (p12r * p12r + p12i * p12i) * 0.999999f;
  if (det != 0.0f) {
    const float fac = 1.0f / det;
    a1r = (p01r * p12r - p01i * p12i - p02r * p11) * fac;
    a1i = (p01i * p12r + p01r * p12i - p02i * p11) * fac;
  }
  if (p11 != 0) {
    const float fac = 1.0f / p11;
    a0r = -(p01r + a1r * p12r + a1i * p12i) * fac;
    a0i = -(p01i + a1i * p12r - a1r * p12i) * fac;
  }
  if (a0r * a0r + a0i * a0i >= 16.0f || a1r * a1r + a1i * a1i >= 16.0f) a0r = a0i = a1r = a1i = 0.0f;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define XE_NOINLINE __attribute__((noinline)) /* rare paths stay out of the hot kernel's register budget */
#else
#define XE_NOINLINE
#endif
/* sbrdec_lpfuncs.c:1251-1352, harmonic patching: every high band is the transposer's band, inverse-filtered with its own
   prediction coefficients */
XE_NOINLINE FX_HD void xe_harmonic_patch(const XsCx cx, const xaac_sbr_header *h, xaac_esbr_state *st, XeWork *w,
                                         const XeMat dst, const XeMat ph, int start, int end, int usb, int num_if) {
  /* (cx, dst, ph by value: a reference to a caller's object would put it on the stack of this out-of-line call) */
  /* sbrdec_lpfuncs.c:1251-1344: every high band is the transposer's band, inverse-filtered with its own prediction
     coefficients; the chirp factor by the noise-floor band the band lies in (a running index in the reference, which
     gives up at the table's fifth entry) */
  cx.sync();
  if (w->err) return;
  XS_PAR(k2, h->sub_band_start, usb) {
    float c0r, c0i, c1r, c1i;
    xe_covar_alpha(ph, k2, c0r, c0i, c1r, c1i);
    int bw_index = 0;
    while (bw_index < 5 && k2 >= h->freq_band_tbl_noise[1 + bw_index]) bw_index++;
    if (bw_index >= 5) {
      w->err = -1;
    } else {
      float bw = w->bw_array[bw_index];
      const float a0r = bw * c0r, a0i = bw * c0i;
      bw *= bw;
      const float a1r = bw * c1r, a1i = bw * c1i;
      float r2 = ph.r(start - 2, k2), i2 = ph.i(start - 2, k2), r1 = ph.r(start - 1, k2), i1 = ph.i(start - 1, k2);
      XE_NOUNROLL
      for (int l0 = start; l0 < end; l0 += XE_CH) {
        float cr[XE_CH] = {0}, ci[XE_CH] = {0};
        xe_rows_load(ph, k2, l0, end, cr, ci);
        XE_UNROLL
        for (int j = 0; j < XE_CH; j++)
          if (l0 + j < end) {
            const float r0 = cr[j], i0 = ci[j];
            float yr = r0, yi = i0;
            if (bw > 0.0f) {
              yr += ((a0r * r1 - a0i * i1) + (a1r * r2 - a1i * i2));
              yi += ((a0i * r1 + a0r * i1) + (a1i * r2 + a1r * i2));
            }
            cr[j] = yr;
            ci[j] = yi;
            r2 = r1; i2 = i1;
            r1 = r0; i1 = i0;
          }
        xe_rows_store(dst, k2, l0, end, cr, ci);
      }
    }
  }
  cx.sync();
  if (w->err) return;
  XS_ONE st->num_patches = 1; /* :1345-1352: patch = 1 on this path */
  XS_PAR(i, 0, num_if) st->bw_array_prev[i] = w->bw_array[i];
  cx.sync();
  return;
}

/* ixheaacd_pre_processing (sbrdec_lpfuncs.c:928-979): the pre-flattening of LPP patches a frame's ENHSBR element asks for
   (bs_sbr_preprocessing; env_extr.c:602).  The low band's envelope in dB per QMF band over the frame's slots, a cubic fitted
   to it over the band index (ixheaacd_polyfit :898, normal equations solved by ixheaacd_gausssolve :850 with partial
   pivoting -- all in single precision, restated operation by operation), and per band the gain that brings the fitted
   curve to the mean level.  Gains -> w->nrg_gain[0 .. num_bands); w->nrg_est, w->alpha_r are scratch here. */
XE_NOINLINE FX_HD void xe_pre_flatten(const XsCx cx, XeWork *w, const XeMat src, int num_bands, int start, int end) {
  float *low_env = w->nrg_est, *gain = w->nrg_gain;
  float *a = &w->alpha_r[0][0], *b = a + 16, *v = b + 4, *p = v + 7, *mean = p + 4; /* a[4][4] b[4] v[7] p[4] mean */
  XS_PAR(k, 0, 64) {
    float e = 0.0f;
    if (k < num_bands && num_bands != 0 && end != start) {
      float temp = 0.0f;
      for (int i = start; i < end; i++) temp += src.r(i, k) * src.r(i, k) + src.i(i, k) * src.i(i, k);
      temp /= (float)(end - start);
      e = xm_10log10f_of(temp + 1);
    }
    low_env[k] = e;
  }
  cx.sync();
  XS_ONE {
    float m = 0.0f;
    if (num_bands != 0 && end != start) {
      for (int k = 0; k < num_bands; k++) m = m + low_env[k];
      m /= (float)num_bands;
    }
    *mean = m;
    for (int i = 0; i < 20; i++) a[i] = 0.0f; /* a and b */
    for (int k = 0; k < num_bands; k++) {
      v[0] = 1.0f;
      for (int i = 1; i <= 6; i++) v[i] = (float)k * v[i - 1];
      for (int i = 0; i <= 3; i++) {
        b[i] += v[3 - i] * low_env[k];
        for (int j = 0; j <= 3; j++) a[4 * i + j] += v[6 - i - j];
      }
    }
    for (int i = 0; i < 4; i++) {
      int imax = i;
      for (int k = i + 1; k < 4; k++)
        if (fabsf(a[4 * k + i]) > fabsf(a[4 * imax + i])) imax = k;
      if (imax != i) {
        float t = b[imax];
        b[imax] = b[i];
        b[i] = t;
        for (int j = i; j < 4; j++) {
          t = a[4 * imax + j];
          a[4 * imax + j] = a[4 * i + j];
          a[4 * i + j] = t;
        }
      }
      const float d = a[4 * i + i];
      b[i] /= d;
      for (int j = i; j < 4; j++) a[4 * i + j] /= d;
      for (int k = i + 1; k < 4; k++) {
        const float t = a[4 * k + i];
        b[k] -= t * b[i];
        for (int j = i + 1; j < 4; j++) a[4 * k + j] -= t * a[4 * i + j];
      }
    }
    for (int i = 3; i >= 0; i--) {
      p[i] = b[i];
      for (int j = i + 1; j < 4; j++) p[i] -= a[4 * i + j] * p[j];
    }
  }
  cx.sync();
  XS_PAR(k, 0, 64) {
    if (k < num_bands) {
      float x = (float)k;
      float slope = p[3];
      slope = slope + p[2] * x;
      x = x * x;
      slope = slope + p[1] * x;
      x = x * (float)k;
      slope = slope + p[0] * x;
      gain[k] = xm_pow10f_of((*mean - slope) / 20.0f);
    }
  }
  cx.sync();
}

/* HARM = false: a build without the harmonic branch (the kernel variant for batches without transposers: a frame with
   harmonic_sbr set is refused there) */
template <bool HARM = true>
FX_HD void xe_generate_hf(const XsCx &cx, const xaac_sbr_header *h, const xaac_sbr_frame *f, const xaac_esbr_side *sd,
                          xaac_esbr_state *st, XeWork *w, const XeMat &src, const XeMat &dst,
                          const XeMat *ph = nullptr /* ph_vocod_qmf rows (row 0 = the reference's + 2), harmonic patching */,
                          int rate = 2 /* QMF slots per envelope time slot: 4 for 4:1 SBR (is_usf_4, :1036-1044) */) {
  const int start = rate * f->border_vec[0], end = 16 * rate + rate * (f->border_vec[f->num_env] - 16);
  const int lsb = sd->f_master_tbl[0], usb = sd->f_master_tbl[sd->num_mf_bands];
  const int num_if = h->num_nf_bands;
  XS_PAR(i, 0, XAAC_SBR_MAX_PATCHES) { /* chirp factors, :832 */
    float bw = 0.0f;
    if (i < num_if) {
      const float tab[4][4] = {{0.00f, 0.60f, 0.90f, 0.98f}, {0.60f, 0.75f, 0.90f, 0.98f},
                                      {0.00f, 0.75f, 0.90f, 0.98f}, {0.00f, 0.75f, 0.90f, 0.98f}}; /* :80 */
      bw = tab[sd->sbr_invf_mode_prev[i] & 3][f->sbr_invf_mode[i] & 3];
      if (bw < st->bw_array_prev[i]) bw = 0.75000f * bw + 0.25000f * st->bw_array_prev[i];
      else bw = 0.90625f * bw + 0.09375f * st->bw_array_prev[i];
      if (bw < 0.015625f) bw = 0;
    }
    w->bw_array[i] = bw;
  }
  /* sbrdec_lpfuncs.c:1065 / :1252: harmonic patches need sbr_patching_mode 0 AND a transposer (hbe_flag); a USAC channel without
     one (XAAC_ESBR_NO_X_DELAY) carries sbr_patching_mode 0 in every FD frame and takes the LPP patches */
  const bool harmonic = (sd->harmonic_sbr & XAAC_ESBR_HARMONIC) && !(sd->harmonic_sbr & XAAC_ESBR_NO_X_DELAY);
  XS_ONE {
    w->err = 0;
    if (!harmonic) xe_build_patches(h, sd, st, w);
    else if (!HARM || !ph) w->err = -1; /* no transposer behind this channel */
  }
  XS_PAR(k, usb, 64)
    for (int l = start; l < end; l++) {
      dst.r(l, k) = 0.0f;
      dst.i(l, k) = 0.0f;
    }
  if constexpr (HARM) {
    if (harmonic) {
      cx.sync();
      if (w->err) return;
      xe_harmonic_patch(cx, h, st, w, dst, *ph, start, end, usb, num_if);
      return;
    }
  }
  const bool flatten = (sd->harmonic_sbr & XAAC_ESBR_PRE_FLATTEN) != 0;
  if (flatten) {
    cx.sync();
    xe_pre_flatten(cx, w, src, lsb, start, end); /* (before the prediction coefficients, whose arrays it borrows) */
  }
  XS_PAR(k, 0, 64) {
    float a0r = 0, a0i = 0, a1r = 0, a1i = 0;
    if (k >= 1 && k < lsb) xe_covar_alpha(src, k, a0r, a0i, a1r, a1i, rate == 4 ? 76 : 38);
    w->alpha_r[k][0] = a0r; w->alpha_i[k][0] = a0i;
    w->alpha_r[k][1] = a1r; w->alpha_i[k][1] = a1i;
  }
  cx.sync();
  if (w->err) return;
  xe_patch_bw_index(cx, h, w);
  cx.sync();
  if (w->err) return;
  XS_PAR(k2, 0, 64) {
    const int k = w->src_band[k2];
    if (k == -1 || (k == -2 && k2 >= h->sub_band_start && k2 < usb)) { /* (a band the patch loop left out keeps stale data
                                                                         in the reference; here it is silent) */
      for (int l = start; l < end; l++) {
        dst.r(l, k2) = 0.0f;
        dst.i(l, k2) = 0.0f;
      }
    } else if (k >= 0) {
      float bw = w->bw_array[w->bw_idx[k2]];
      const float a0r = bw * w->alpha_r[k][0], a0i = bw * w->alpha_i[k][0];
      bw *= bw;
      const float a1r = bw * w->alpha_r[k][1], a1i = bw * w->alpha_i[k][1];
      float r2 = src.r(start - 2, k), i2 = src.i(start - 2, k), r1 = src.r(start - 1, k), i1 = src.i(start - 1, k);
      /* :1220-1247; the gain of a pre-flattened frame as its own copy of the loop: the usual frame multiplies by
         nothing (x * 1.0f = x exactly, so leaving the products out changes no value) */
      const auto run = [&](auto fl) {
        const float gain = decltype(fl)::value ? w->nrg_gain[k] : 1.0f;
        XE_NOUNROLL
        for (int l0 = start; l0 < end; l0 += XE_CH) {
          float cr[XE_CH] = {0}, ci[XE_CH] = {0};
          xe_rows_load(src, k, l0, end, cr, ci);
          XE_UNROLL
          for (int j = 0; j < XE_CH; j++)
            if (l0 + j < end) {
              const float r0 = cr[j], i0 = ci[j];
              float yr = r0 * gain, yi = i0 * gain;
              if (bw > 0.0f) {
                yr += (a0r * r1 - a0i * i1 + a1r * r2 - a1i * i2) * gain;
                yi += (a0i * r1 + a0r * i1 + a1i * r2 + a1r * i2) * gain;
              }
              cr[j] = yr;
              ci[j] = yi;
              r2 = r1; i2 = i1;
              r1 = r0; i1 = i0;
            }
          xe_rows_store(dst, k2, l0, end, cr, ci);
        }
      };
      if (flatten) run(XeTrue{});
      else run(XeFalse{});
    }
  }
  XS_PAR(i, 0, num_if) st->bw_array_prev[i] = w->bw_array[i];
  cx.sync();
}

/* sbr_dec.c:866-872: with the harmonic-transposer flag set -- which the reference sets for every non-USAC stream it
   decodes with eSBR on (sbrdecoder.c:399-403) -- rows 2..7 of the shifted qmf_buf lose what lies above the previous
   frame's cross-over band.  The reference's memset counts BYTES where it means floats: (64 - qmf_sb_prev) bytes from band
   qmf_sb_prev on, i.e. a quarter of the bands, and for a count that is not a multiple of four the low bytes of one more
   float.  Restated to the byte.  (Nothing reads these cells unless the cross-over band moves up between frames.) */
FX_HD void xe_hbe_history_clear(const XsCx &cx, xaac_esbr_state *st, int qmf_sb_prev) {
  const int bytes = 64 - qmf_sb_prev, full = bytes >> 2, rem = bytes & 3;
  XS_PAR(k, 0, 64) {
    if (k >= qmf_sb_prev && k < qmf_sb_prev + full + (rem ? 1 : 0)) {
      const uint32_t keep = k < qmf_sb_prev + full ? 0u : ~((1u << (8 * rem)) - 1u);
      uint32_t va[6], vb[6]; /* the twelve words in flight together (as twelve read-modify-writes through pointers the
                                compiler cannot tell apart they were twelve memory round trips in a row) */
      XE_UNROLL
      for (int r = 0; r < 6; r++) {
        memcpy(&va[r], &st->qmf_re[2 + r][k], 4);
        memcpy(&vb[r], &st->qmf_im[2 + r][k], 4);
      }
      XE_UNROLL
      for (int r = 0; r < 6; r++) {
        va[r] &= keep;
        vb[r] &= keep;
        memcpy(&st->qmf_re[2 + r][k], &va[r], 4);
        memcpy(&st->qmf_im[2 + r][k], &vb[r], 4);
      }
    }
  }
  cx.sync();
}

/* ---- ixheaacd_apply_inter_tes (gamma > 0 only; HE-AAC streams carry gamma 0) ---------------------------------------- */
FX_HD void xe_inter_tes(const XsCx &cx, XeWork *w, const XeMat &low, const XeMat &x, int row0, int num_sample,
                        int sb_start, int num_sb, int gamma_idx) {
  const float gamma = xaac_esbr_q_gamma[gamma_idx & 3];
  if (!(gamma > 0)) return;
  const int sb_end = sb_start + num_sb;
  XS_PAR(i, 0, num_sample) {
    const int l = row0 + i;
    for (int j = 0; j < sb_start; j++) {
      x.r(l, j) = low.r(l, j);
      x.i(l, j) = low.i(l, j);
    }
    float pl = 0.0f, ph = 0.0f;
    for (int j = 0; j < sb_start; j++) {
      pl += x.r(l, j) * x.r(l, j);
      pl += x.i(l, j) * x.i(l, j);
    }
    for (int j = sb_start; j < sb_end; j++) {
      ph += x.r(l, j) * x.r(l, j);
      ph += x.i(l, j) * x.i(l, j);
    }
    w->pow_lo[i] = pl;
    w->pow_hi[i] = ph;
  }
  cx.sync();
  float tot_lo = 0.0f, tot_hi = 0.0f, tot_after = 1.0e-6f;
  for (int i = 0; i < num_sample; i++) {
    tot_lo += w->pow_lo[i];
    tot_hi += w->pow_hi[i];
  }
  XS_PAR(i, 0, num_sample) {
    float g = (float)xe_sqrt((double)(w->pow_lo[i] * (float)num_sample / (tot_lo + 1.0e-6f)));
    g = (float)(1.0f + gamma * (g - 1.0f));
    if (g < 0.2f) g = 0.2f;
    w->tes_gain[i] = g;
  }
  cx.sync();
  for (int i = 0; i < num_sample; i++) {
    const float g = w->tes_gain[i];
    tot_after += w->pow_hi[i] * (g * g);
  }
  const float gain_adj = (float)xe_sqrt((double)(tot_hi / tot_after));
  XS_PAR(k, sb_start, sb_end)
    for (int i = 0; i < num_sample; i++) {
      const float g = w->tes_gain[i] * gain_adj;
      x.r(row0 + i, k) *= g;
      x.i(row0 + i, k) *= g;
    }
  cx.sync();
}

/* ---- ixheaacd_sbr_env_calc, PVC_SBR (esbr_envcal.c:194-607) -----------------------------------------------------------
   The envelope is the PVC decoder's: one reference energy per band and PVC time slot (= two QMF slots), so gains, limiter and
   boost are made per time slot; the reference makes them for all of an envelope's slots first and applies them afterwards -- a
   slot's gains read nothing an earlier slot's application writes (rows 2 t, 2 t + 1 of sbr_qmf_out only), so here slot t is
   adjusted as soon as its gains exist and the [64][48] work matrices never materialise.  lane = band for the element-wise
   steps, lane = limiter band for the three sums in band order, as in the ORIG_SBR branch.
   The frame grids: the noise floor is mapped to qmapped_pvc with the SBR grid (str_frame_info_details), the envelopes walk the
   PVC grid (pvs->border_vec / freq_res); slots below sin_len_for_cur_top use the previous frame's frequency resolution and
   the sinusoids that frame announced (harm_flag_varlen).  Returns 0 or -1. */
FX_HD int xe_env_calc_pvc(const XsCx &cx, const xaac_sbr_header *h, const xaac_sbr_frame *f, const xaac_esbr_side *sd,
                          xaac_esbr_state *st, XeWork *w, const XeMat &x, const xaac_esbr_pvc_side *pvs, xaac_esbr_pvc_state *pst,
                          const float *env_out, int &phase_index, int &harm_index, int &start_up, int &start_up_pvc, int rate = 2) {
  const int sb_start = h->sub_band_start, num_sb = h->sub_band_end - h->sub_band_start;
  const int num_env = f->num_env, trans_env = f->transient_env, num_nf = h->num_nf_bands;
  const int smoothing_length = h->smoothing_mode ? 0 : 4, int_mode = h->interpol_freq;
  const int lim_band = sd->limiter_bands & 3, lim_gains = h->limiter_gains & 3;
  const double guard = 1e-17;
  if (!env_out) return -1;
  /* grids the reference indexes with: its own checks (esbr_envcal.c:234, :283) and the [16][64] envelope's extent */
  for (int i = 0; i < num_env; i++) {
    if (f->border_vec[i] < 0 || f->border_vec[i + 1] > XAAC_ESBR_PVC_COLS) return -1;
    if (pvs->border_vec[i] < 0 || pvs->border_vec[i + 1] > XAAC_PVC_SLOTS) return -1;
    if ((unsigned)pvs->freq_res[i] > 1u) return -1;
  }
  /* (the reference's first loop over an envelope's slots, :299, runs on to sin_len_for_cur_top even past the envelope's end;
     what it makes there is made again by the next envelope's loops before anything reads it, so each envelope's own slots are
     all that counts) */
  if ((unsigned)pst->var_len_id_prev > 1u || (unsigned)pst->prev_freq_res[pst->var_len_id_prev] > 1u) return -1;
  /* :210-214: the slots in front of this frame's first border are what the frame before mapped beyond its sixteenth */
  XS_PAR(c, 0, 64)
    for (int t = 0; t < f->border_vec[0]; t++) pst->qmapped[c][t] = pst->qmapped[c][t + 16];
  cx.sync();
  { /* :216-262: the noise floor of every band and slot, on the SBR grid */
    int kk = 0, next = -1;
    for (int i = 0; i < num_env; i++) {
      if (kk > XAAC_SBR_MAX_NOISE_ENVELOPES) return -1;
      if (f->border_vec[i] == f->noise_border_vec[kk]) kk++, next++;
      if (next < 0) return -1; /* the reference would read in front of flt_noise_floor */
      const int res = f->freq_res[i] ? 1 : 0, nsf = h->num_sf_bands[res];
      const int16_t *ftab = res ? h->freq_band_tbl_hi : h->freq_band_tbl_lo;
      const int n_bands = ftab[nsf] - ftab[0];
      XS_PAR(c, 0, n_bands < 64 ? n_bands : 64) {
        const int kabs = ftab[0] + c;
        int o = 0;
        for (int q = 1; q < num_nf; q++) o += kabs >= h->freq_band_tbl_noise[q];
        const float nf = sd->flt_noise_floor[next * num_nf + o];
        for (int t = f->border_vec[i]; t < f->border_vec[i + 1]; t++) pst->qmapped[c][t] = nf;
      }
    }
  }
  cx.sync();
  int kk = 0, next = -1;
  for (int i = 0; i < num_env; i++) {
    if (kk > XAAC_SBR_MAX_NOISE_ENVELOPES) return -1;
    if (f->border_vec[i] == f->noise_border_vec[kk]) kk++, next++;
    const int start_pos = pvs->border_vec[i], end_pos = pvs->border_vec[i + 1];
    int noise_absc = (i == trans_env || i == st->env_short_flag_prev) ? 1 : 0;
    if (pst->prev_sbr_mode == XAAC_ESBR_SBR_ORIG) noise_absc = 0; /* :294 */
    const int smooth_length = noise_absc ? 0 : smoothing_length;
    const float *filt = smooth_length ? xaac_esbr_fir_4 : xaac_esbr_fir_0;
    for (int t = start_pos; t < end_pos; t++) {
      const bool tail = t < pvs->sin_len_for_cur_top; /* :299 / :432: the two loops over an envelope's slots */
      const int res = (tail ? pst->prev_freq_res[pst->var_len_id_prev] : pvs->freq_res[i]) ? 1 : 0, nsf = h->num_sf_bands[res];
      const int16_t *ftab = res ? h->freq_band_tbl_hi : h->freq_band_tbl_lo;
      /* which bands' sinusoids count in this slot: the old frame's announcement in its tail, this frame's behind it */
      XS_PAR(cc, 0, 64)
        w->own_tone[cc] = (int8_t)(tail ? (pst->harm_flag_varlen[cc] && (t >= pvs->sin_start_for_cur_top || pst->harm_flag_varlen_prev[(cc + sb_start) & 63]))
                                        : (w->harmonics[cc] && (t >= pvs->sine_position || w->harm_prev[(cc + sb_start) & 63])));
      cx.sync();
      XS_PAR(c, 0, num_sb) {
        const int kabs = sb_start + c;
        int j = 0, o = 0;
        for (int q = 1; q < nsf; q++) j += kabs >= ftab[q];
        const int li = ftab[j], ui = ftab[j + 1];
        int flag = 0;
        for (int k = li; k < ui; k++) flag |= w->own_tone[(k - sb_start) & 63];
        for (int q = 1; q < num_nf; q++) o += kabs >= h->freq_band_tbl_noise[q];
        w->sfb_first[c] = (int8_t)(li - sb_start);
        w->sfb_len[c] = (int8_t)(ui - li);
        w->flag[c] = (int8_t)flag;
        w->o_idx[c] = (int8_t)o;
        float nrg = 0; /* :322-327: the slot's QMF rows (two; four for 4:1 SBR) */
        for (int l = 0; l < rate; l++) nrg += (x.r(rate * t + l, kabs) * x.r(rate * t + l, kabs)) + (x.i(rate * t + l, kabs) * x.i(rate * t + l, kabs));
        w->nrg_est[c] = nrg / rate;
      }
      cx.sync();
      XS_PAR(c, 0, num_sb) {
        float est = w->nrg_est[c];
        if (!int_mode) {
          float nrg = 0;
          const int n = w->sfb_len[c], c0 = w->sfb_first[c];
          for (int k = c0; k < c0 + n; k++) nrg += w->nrg_est[k];
          est = nrg / (float)n;
        }
        const float ref = env_out[64 * t + sb_start + c];
        const float q = pst->qmapped[c][t];
        const double tmp = q / (1 + q + guard);
        float gain, tone = 0;
        if (w->flag[c]) {
          gain = (float)xe_sqrt(ref * tmp / (est + 1));
          /* :360-367: this frame's own sinusoid ... */
          if (w->harmonics[c] && (t >= pvs->sine_position || w->harm_prev[(c + sb_start) & 63])) tone = (float)xe_sqrt(ref * tmp / (q + guard));
          if (tail) { /* :369-375: ... and, in the tail, the old frame's, over the old noise level */
            if (pst->harm_flag_varlen[c] && (t >= pvs->sin_start_for_cur_top || pst->harm_flag_varlen_prev[(c + sb_start) & 63]))
              tone = (float)xe_sqrt(ref * tmp / (pst->prev_noise_level[w->o_idx[c]] + guard));
          }
        } else if (noise_absc) {
          gain = (float)xe_sqrt(ref / (est + 1));
        } else {
          gain = (float)xe_sqrt(ref * tmp / ((est + 1) * (q + guard)));
        }
        w->nrg_ref[c] = ref;
        w->nrg_gain[c] = gain;
        w->nrg_tone[c] = tone;
        w->noise_level[c] = (float)xe_sqrt(ref * tmp);
        w->pow_lo[c] = est;
      }
      cx.sync();
      XS_PAR(c, 0, num_sb) w->nrg_est[c] = w->pow_lo[c];
      cx.sync();
      /* the limiter (:390-428 / :513-551), as in the ORIG_SBR branch */
      const int n_lim = st->gate_mode[lim_band] < 12 ? st->gate_mode[lim_band] : 12;
      XS_PAR(c, 0, n_lim) {
        const int k0 = w->lim_tab[c], k1r = w->lim_tab[c + 1], k1 = k1r < num_sb ? k1r : num_sb;
        float p_ref = 0, p_est = 0, g_max = 0;
        if (k0 >= 0 && k0 <= k1) {
          for (int k = k0; k < k1; k++) {
            p_ref += w->nrg_ref[k];
            p_est += w->nrg_est[k];
          }
          const float avg_gain = (float)xe_sqrt((p_ref + 1e-12f) / (p_est + 1e-12f));
          g_max = avg_gain * xaac_esbr_g_lim_gains[lim_gains];
          if (g_max > 1.0e5f) g_max = 1.0e5f;
        }
        w->lim_pref[c] = p_ref;
        w->lim_gmax[c] = g_max;
      }
      cx.sync();
      XS_PAR(k, 0, num_sb) {
        int lb = -1;
        for (int c = 0; c < n_lim; c++) {
          const int k0 = w->lim_tab[c], k1r = w->lim_tab[c + 1], k1 = k1r < num_sb ? k1r : num_sb;
          if (lb < 0 && k0 >= 0 && k >= k0 && k < k1) lb = c;
        }
        w->lim_of[k] = (int8_t)lb;
        float ta = 0.0f, tb = 0.0f;
        if (lb >= 0) {
          const float g_max = w->lim_gmax[lb];
          if (g_max <= w->nrg_gain[k]) {
            w->noise_level[k] = (float)(w->noise_level[k] * (g_max / (w->nrg_gain[k] + guard)));
            w->nrg_gain[k] = g_max;
          }
          ta = w->nrg_gain[k] * w->nrg_gain[k] * w->nrg_est[k];
          if (w->nrg_tone[k]) tb = w->nrg_tone[k] * w->nrg_tone[k];
          else if (!noise_absc) tb = w->noise_level[k] * w->noise_level[k];
        }
        w->pow_hi[k] = ta;
        w->tes_gain[k] = tb;
      }
      cx.sync();
      XS_PAR(c, 0, n_lim) {
        const int k0 = w->lim_tab[c], k1r = w->lim_tab[c + 1], k1 = k1r < num_sb ? k1r : num_sb;
        float boost = 1.0f;
        if (k0 >= 0 && k0 <= k1) {
          float p_adj = 0;
          for (int k = k0; k < k1; k++) {
            p_adj += w->pow_hi[k];
            p_adj += w->tes_gain[k];
          }
          boost = (float)xe_sqrt((w->lim_pref[c] + 1e-12f) / (p_adj + 1e-12f));
          boost = boost > 1.584893192f ? 1.584893192f : boost;
        }
        w->lim_gmax[c] = boost;
      }
      cx.sync();
      XS_PAR(k, 0, num_sb) {
        const int lb = w->lim_of[k];
        if (lb >= 0) {
          const float boost = w->lim_gmax[lb];
          w->nrg_gain[k] *= boost;
          w->noise_level[k] *= boost;
          w->nrg_tone[k] *= boost;
        }
      }
      cx.sync();
      if (t == start_pos && start_up_pvc) { /* :554-563: the smoothing histories start from the envelope's first slot */
        XS_PAR(k, 0, num_sb)
          for (int n = 0; n < 4; n++) {
            st->e_gain[n][k] = w->nrg_gain[k];
            st->noise_buf[n][k] = w->noise_level[k];
          }
        start_up_pvc = 0;
        start_up = 0;
      }
      cx.sync();
      XS_PAR(k, 0, 64) { /* :564-606: the slot's two QMF rows; the five-deep histories rotate once per row for all 64 bands */
        float eg[5], nb[5];
        for (int n = 0; n < 5; n++) {
          eg[n] = st->e_gain[n][k];
          nb[n] = st->noise_buf[n][k];
        }
        const bool active = k < num_sb;
        const float gain = active ? w->nrg_gain[k] : 0.0f, nl = active ? w->noise_level[k] : 0.0f, tone = active ? w->nrg_tone[k] : 0.0f;
        const bool no_noise = active && (tone != 0 || noise_absc);
        const int kk2 = sb_start + k, freq_inv = (kk2 & 1) ? -1 : 1;
        const float hp[2][4] = {{1.0f, 0.0f, -1.0f, 0.0f}, {0.0f, 1.0f, 0.0f, -1.0f}};
        for (int j = 0; j < rate; j++) {
          if (active) {
            eg[4] = gain;
            nb[4] = nl;
            float sb_gain = 0, sb_noise = 0;
            int c = 0;
            for (int n = 4 - smooth_length; n <= 4; n++) {
              sb_gain += eg[n] * filt[c];
              sb_noise += nb[n] * filt[c++];
            }
            const int ph = (phase_index + j * num_sb + k + 1) & 511, hi = (harm_index + j) & 3;
            if (no_noise) sb_noise = 0;
            const float re = x.r(rate * t + j, kk2), im = x.i(rate * t + j, kk2);
            x.r(rate * t + j, kk2) = re * sb_gain + sb_noise * XE_RANDOM_PHASE(2 * ph) + tone * hp[0][hi];
            x.i(rate * t + j, kk2) = im * sb_gain + sb_noise * XE_RANDOM_PHASE(2 * ph + 1) + tone * (float)freq_inv * hp[1][hi];
          }
          const float t0 = eg[0], t1 = nb[0];
          for (int n = 0; n < 4; n++) {
            eg[n] = eg[n + 1];
            nb[n] = nb[n + 1];
          }
          eg[4] = t0;
          nb[4] = t1;
        }
        for (int n = 0; n < 5; n++) {
          st->e_gain[n][k] = eg[n];
          st->noise_buf[n][k] = nb[n];
        }
      }
      phase_index = (phase_index + rate * num_sb) & 511;
      harm_index = (harm_index + rate) & 3;
      cx.sync();
    }
  }
  return 0;
}

/* ---- ixheaacd_sbr_env_calc, ORIG_SBR ------------------------------------------------------------------------------- */
/* pvs / pst / env_out: the PVC side info, state and the PVC decoder's envelope [16][64] of a channel whose host tracks them
   (xaac_esbr.h), or NULL: a frame with pvs->sbr_mode == PVC takes xe_env_calc_pvc's envelopes, and every frame leaves the
   bookkeeping a PVC frame behind it reads (esbr_envcal.c:861-899) */
FX_HD int xe_env_calc(const XsCx &cx, const xaac_sbr_header *h, const xaac_sbr_frame *f, const xaac_esbr_side *sd,
                      xaac_esbr_state *st, XeWork *w, const XeMat &x /* sbr_qmf_out */, const XeMat &low /* qmf_buf */,
                      const int32_t *x_over_qmf = nullptr /* the transposer's, where the host tracks one */,
                      const xaac_esbr_pvc_side *pvs = nullptr, xaac_esbr_pvc_state *pst = nullptr, const float *env_out = nullptr,
                      int rate = 2 /* 4 for 4:1 SBR (esbr_envcal.c:152) */) {
  const int sb_start = h->sub_band_start, num_sb = h->sub_band_end - h->sub_band_start;
  const int num_env = f->num_env, trans_env = f->transient_env, num_nf = h->num_nf_bands;
  const int smoothing_length = h->smoothing_mode ? 0 : 4, int_mode = h->interpol_freq;
  const int lim_band = sd->limiter_bands & 3, lim_gains = h->limiter_gains & 3;
  const double guard = 1e-17;
  int phase_index = st->phase_index, harm_index = st->harm_index, start_up = st->esbr_start_up;
  int start_up_pvc = pst ? pst->esbr_start_up_pvc : 0;
  if (num_sb < 0 || num_sb > 64) return -1;
  if (sd->reset_flag) {
    start_up = 1;
    start_up_pvc = 1;
    phase_index = 0;
    XS_ONE w->err = xe_limiter_bands(h, st, (sd->harmonic_sbr & XAAC_ESBR_HARMONIC) != 0, x_over_qmf, reinterpret_cast<int32_t *>(w->pow_lo));
    cx.sync();
    if (w->err) return -1;
  }
  {
    const int mode = (sd->harmonic_sbr & XAAC_ESBR_HARMONIC) ? 0 : 1; /* sbr_patching_mode; esbr_envcal.c:181-190 */
    const int changed = mode != st->prev_sbr_patching_mode;
    cx.sync();
    if (changed) {
      XS_ONE {
        w->err = xe_limiter_bands(h, st, (sd->harmonic_sbr & XAAC_ESBR_HARMONIC) != 0, x_over_qmf, reinterpret_cast<int32_t *>(w->pow_lo));
        if (!w->err) st->prev_sbr_patching_mode = mode;
      }
      cx.sync();
      if (w->err) return -1;
    }
  }
  XS_PAR(k, 0, 64) w->harm_prev[k] = st->harm_flag_prev[k];
  XS_PAR(c, 0, 13) w->lim_tab[c] = st->lim_table[lim_band][c]; /* (as the reset above may just have made it) */
  /* the sinusoids' bands (esbr_envcal.c:640-655): a scale-factor band's flag goes to its centre band; the centres of a strictly
     increasing table are distinct, so the bands scatter their flags side by side (on lane 0 this loop was fifty LDS round
     trips one behind the other; a centre outside 0..63 fails the frame either way) */
  XS_PAR(i, 0, 64) w->harmonics[i] = 0;
  XS_ONE w->err = 0;
  cx.sync();
  XS_PAR(i, 0, h->num_sf_bands[1]) {
    const int li = h->freq_band_tbl_hi[i], ui = h->freq_band_tbl_hi[i + 1];
    const int tmp = ((ui + li) - (sb_start << 1)) >> 1;
    if (tmp >= 64 || tmp < 0) w->err = -1;
    else w->harmonics[tmp] = (int8_t)f->add_harmonics[i];
  }
  cx.sync();
  if (w->err) return -1;
  const bool pvc_frame = pvs && pst && pvs->sbr_mode == XAAC_ESBR_SBR_PVC;
  if (pvc_frame) {
    if (xe_env_calc_pvc(cx, h, f, sd, st, w, x, pvs, pst, env_out, phase_index, harm_index, start_up, start_up_pvc, rate)) return -1;
  }
  int kk = 0, next = -1, m = 0;
  /* a frame whose sbr_mode is not ORIG_SBR (a USAC channel's first frames: UNKNOWN_SBR) passes the envelopes by: all of an
     envelope's work sits inside `if (sbr_mode == ORIG_SBR)` (esbr_envcal.c:646-857); the reset above and the bookkeeping below run */
  const bool skip_adjust = (sd->harmonic_sbr & XAAC_ESBR_SKIP_ADJUST) != 0;
  for (int i = 0; i < (pvc_frame ? 0 : num_env); i++) {
    if (kk > XAAC_SBR_MAX_NOISE_ENVELOPES) return -1;
    if (f->border_vec[i] == f->noise_border_vec[kk]) kk++, next++;
    if (skip_adjust) continue;
    if (next < 0) return -1; /* the reference would read in front of flt_noise_floor */
    const int noise_absc = (i == trans_env || i == st->env_short_flag_prev) ? 1 : 0;
    const int smooth_length = noise_absc ? 0 : smoothing_length;
    const float *filt = smooth_length ? xaac_esbr_fir_4 : xaac_esbr_fir_0;
    const int res = f->freq_res[i] ? 1 : 0, nsf = h->num_sf_bands[res];
    const int16_t *ftab = res ? h->freq_band_tbl_hi : h->freq_band_tbl_lo;
    const int l0 = rate * f->border_vec[i], l1 = rate * f->border_vec[i + 1];
    /* band -> scale-factor band / noise band map of this envelope (esbr_envcal.c:657-699).  The reference walks the bands
       once, stepping its noise-band counter whenever a band reaches the next noise border; with the strictly increasing
       tables xe_side_info_bad insists on, that counter is the number of inner noise borders at or below the band, so
       every lane finds its own band's entries. */
    XS_PAR(cc, 0, 64) w->own_tone[cc] = (int8_t)(w->harmonics[cc] && (i >= trans_env || w->harm_prev[(cc + sb_start) & 63]));
    cx.sync();
    XS_PAR(c, 0, num_sb) {
      const int kabs = sb_start + c;
      int j = 0; /* (a count over the increasing table instead of a search: the loads do not wait for one another) */
      for (int q = 1; q < nsf; q++) j += kabs >= ftab[q];
      const int li = ftab[j], ui = ftab[j + 1];
      int flag = 0, o = 0;
      for (int k = li; k < ui; k++) flag |= w->own_tone[(k - sb_start) & 63];
      for (int q = 1; q < num_nf; q++) o += kabs >= h->freq_band_tbl_noise[q];
      w->sfb_first[c] = (int8_t)(li - sb_start);
      w->sfb_len[c] = (int8_t)(ui - li);
      w->flag[c] = (int8_t)flag;
      w->o_idx[c] = (int8_t)o;
      w->m_idx[c] = (int16_t)(m + j);
    }
    cx.sync();
    XE_T(6);
    if (w->err) return -1;
    m += nsf;
    XS_PAR(c, 0, num_sb) { /* band energies */
      float nrg = 0;
      const int k = sb_start + c;
      if (l0 < l1) {
        XE_NOUNROLL
        for (int j0 = l0; j0 < l1; j0 += XE_CH) {
          float cr[XE_CH] = {0}, ci[XE_CH] = {0};
          xe_rows_load(x, k, j0, l1, cr, ci);
          XE_UNROLL
          for (int j = 0; j < XE_CH; j++)
            if (j0 + j < l1) nrg += (cr[j] * cr[j]) + (ci[j] * ci[j]);
        }
        nrg = nrg / (float)(l1 - l0);
      }
      w->nrg_est[c] = nrg;
    }
    cx.sync();
    XE_T(7);
    XS_PAR(c, 0, num_sb) { /* gains, :690-722 */
      float est = w->nrg_est[c];
      if (!int_mode) {
        float nrg = 0;
        const int n = w->sfb_len[c], c0 = w->sfb_first[c];
        for (int k = c0; k < c0 + n; k++) nrg += w->nrg_est[k];
        est = nrg / (float)n;
      }
      const float ref = sd->flt_env_sf_arr[w->m_idx[c]];
      const float nf = sd->flt_noise_floor[next * num_nf + w->o_idx[c]];
      const double tmp = nf / (1 + nf + guard);
      const bool tone_here = w->harmonics[c] && (i >= trans_env || w->harm_prev[(c + sb_start) & 63]);
      float gain, tone = 0;
      if (w->flag[c]) {
        gain = (float)xe_sqrt(ref * tmp / (est + 1));
        if (tone_here) tone = (float)xe_sqrt(ref * tmp / fabs(nf + guard));
      } else if (noise_absc) {
        gain = (float)xe_sqrt(ref / (est + 1));
      } else {
        gain = (float)xe_sqrt(ref * tmp / ((est + 1) * fabs(nf + guard)));
      }
      w->nrg_ref[c] = ref;
      w->nrg_gain[c] = gain;
      w->nrg_tone[c] = tone;
      w->noise_level[c] = (float)xe_sqrt(ref * tmp);
      w->pow_lo[c] = est; /* the interpolated estimate, used from here on */
    }
    cx.sync();
    XS_PAR(c, 0, num_sb) w->nrg_est[c] = w->pow_lo[c];
    cx.sync();
    XE_T(8);
    /* limiter (:725-761).  Per limiter band the reference runs three sums over its bands in band order -- those stay
       sequential, one limiter band per lane -- with element-wise work between them, which goes to one band per lane.
       (A table made for another header -- a header change without the reset the parser raises with it -- may reach past
       this frame's bands, where the reference reads whatever its scratch holds; here such a band ends at the last band.) */
    const int n_lim = st->gate_mode[lim_band] < 12 ? st->gate_mode[lim_band] : 12;
    XS_PAR(c, 0, n_lim) {
      const int k0 = w->lim_tab[c], k1r = w->lim_tab[c + 1], k1 = k1r < num_sb ? k1r : num_sb;
      float p_ref = 0, p_est = 0, g_max = 0;
      if (k0 >= 0 && k0 <= k1) {
        for (int k = k0; k < k1; k++) {
          p_ref += w->nrg_ref[k];
          p_est += w->nrg_est[k];
        }
        const float avg_gain = (float)xe_sqrt((p_ref + 1e-12f) / (p_est + 1e-12f));
        g_max = avg_gain * xaac_esbr_g_lim_gains[lim_gains];
        if (g_max > 1.0e5f) g_max = 1.0e5f;
      }
      w->lim_pref[c] = p_ref;
      w->lim_gmax[c] = g_max;
    }
    cx.sync();
    XS_PAR(k, 0, num_sb) {
      int lb = -1;
      for (int c = 0; c < n_lim; c++) {
        const int k0 = w->lim_tab[c], k1r = w->lim_tab[c + 1], k1 = k1r < num_sb ? k1r : num_sb;
        if (lb < 0 && k0 >= 0 && k >= k0 && k < k1) lb = c;
      }
      w->lim_of[k] = (int8_t)lb;
      float ta = 0.0f, tb = 0.0f;
      if (lb >= 0) {
        const float g_max = w->lim_gmax[lb];
        if (g_max <= w->nrg_gain[k]) {
          w->noise_level[k] = (float)(w->noise_level[k] * (g_max / (w->nrg_gain[k] + guard)));
          w->nrg_gain[k] = g_max;
        }
        ta = w->nrg_gain[k] * w->nrg_gain[k] * w->nrg_est[k];
        if (w->nrg_tone[k]) tb = w->nrg_tone[k] * w->nrg_tone[k];
        else if (!noise_absc) tb = w->noise_level[k] * w->noise_level[k];
      }
      w->pow_hi[k] = ta;   /* the two terms the band adds to its limiter band's adjusted power, in this order; an absent */
      w->tes_gain[k] = tb; /* second term is +0, which leaves the non-negative running sum as it is */
    }
    cx.sync();
    XS_PAR(c, 0, n_lim) {
      const int k0 = w->lim_tab[c], k1r = w->lim_tab[c + 1], k1 = k1r < num_sb ? k1r : num_sb;
      float boost = 1.0f;
      if (k0 >= 0 && k0 <= k1) {
        float p_adj = 0;
        for (int k = k0; k < k1; k++) {
          p_adj += w->pow_hi[k];
          p_adj += w->tes_gain[k];
        }
        boost = (float)xe_sqrt((w->lim_pref[c] + 1e-12f) / (p_adj + 1e-12f));
        boost = boost > 1.584893192f ? 1.584893192f : boost;
      }
      w->lim_gmax[c] = boost;
    }
    cx.sync();
    XS_PAR(k, 0, num_sb) {
      const int lb = w->lim_of[k];
      if (lb >= 0) {
        const float boost = w->lim_gmax[lb];
        w->nrg_gain[k] *= boost;
        w->noise_level[k] *= boost;
        w->nrg_tone[k] *= boost;
      }
    }
    cx.sync();
    XE_T(9);
    if (start_up) {
      XS_PAR(k, 0, num_sb)
        for (int n = 0; n < 4; n++) {
          st->e_gain[n][k] = w->nrg_gain[k];
          st->noise_buf[n][k] = w->noise_level[k];
        }
      start_up = 0;
      start_up_pvc = 0; /* esbr_envcal.c:768-769 */
    }
    const bool tes = xaac_esbr_q_gamma[sd->inter_temp_shape_mode[i] & 3] > 0; /* inter-TES works across bands: through memory */
    XS_PAR(k, 0, 64) { /* apply: smoothed gain, noise; the two five-deep histories rotate once per slot, :771-817 */
      float eg[5], nb[5];
      for (int n = 0; n < 5; n++) {
        eg[n] = st->e_gain[n][k];
        nb[n] = st->noise_buf[n][k];
      }
      const bool active = k < num_sb;
      const float gain = active ? w->nrg_gain[k] : 0.0f, nl = active ? w->noise_level[k] : 0.0f;
      const bool no_noise = active && (w->nrg_tone[k] != 0 || noise_absc);
      const int kk2 = sb_start + k;
      const float tone = active ? w->nrg_tone[k] : 0.0f;
      const int freq_inv = (kk2 & 1) ? -1 : 1;
      const float hp[2][4] = {{1.0f, 0.0f, -1.0f, 0.0f}, {0.0f, 1.0f, 0.0f, -1.0f}};
      XE_NOUNROLL
      for (int j0 = l0; j0 < l1; j0 += XE_CH) {
        float cr[XE_CH] = {0}, ci[XE_CH] = {0};
        if (active) xe_rows_load(x, kk2, j0, l1, cr, ci);
        XE_UNROLL
        for (int jj = 0; jj < XE_CH; jj++)
          if (j0 + jj < l1) {
            const int j = j0 + jj - l0;
            if (active) {
              eg[4] = gain;
              nb[4] = nl;
              float sb_gain = 0, sb_noise = 0;
              int c = 0;
              for (int n = 4 - smooth_length; n <= 4; n++) {
                sb_gain += eg[n] * filt[c];
                sb_noise += nb[n] * filt[c++];
              }
              const int ph = (phase_index + j * num_sb + k + 1) & 511;
              if (no_noise) sb_noise = 0;
              cr[jj] = cr[jj] * sb_gain + sb_noise * XE_RANDOM_PHASE(2 * ph);
              ci[jj] = ci[jj] * sb_gain + sb_noise * XE_RANDOM_PHASE(2 * ph + 1);
              if (!tes) { /* sinusoids, :833-850 (behind the inter-TES shaping when that is active: below) */
                const int hi = (harm_index + j) & 3;
                cr[jj] += tone * hp[0][hi];
                ci[jj] += tone * (float)freq_inv * hp[1][hi];
              }
            }
            const float t0 = eg[0], t1 = nb[0];
            for (int n = 0; n < 4; n++) {
              eg[n] = eg[n + 1];
              nb[n] = nb[n + 1];
            }
            eg[4] = t0;
            nb[4] = t1;
          }
        if (active) xe_rows_store(x, kk2, j0, l1, cr, ci);
      }
      for (int n = 0; n < 5; n++) {
        st->e_gain[n][k] = eg[n];
        st->noise_buf[n][k] = nb[n];
      }
    }
    phase_index = (phase_index + (l1 > l0 ? (l1 - l0) * num_sb : 0)) & 511;
    cx.sync();
    XE_T(10);
    if (tes) {
      xe_inter_tes(cx, w, low, x, l0, l1 - l0, sb_start, num_sb, sd->inter_temp_shape_mode[i]);
      XS_PAR(k, 0, num_sb) { /* sinusoids, :833-850 */
        const float tone = w->nrg_tone[k];
        const int freq_inv = ((sb_start + k) & 1) ? -1 : 1;
        int hi = harm_index;
        const float hp[2][4] = {{1.0f, 0.0f, -1.0f, 0.0f}, {0.0f, 1.0f, 0.0f, -1.0f}};
        for (int l = l0; l < l1; l++) {
          x.r(l, sb_start + k) += tone * hp[0][hi];
          x.i(l, sb_start + k) += tone * (float)freq_inv * hp[1][hi];
          hi = (hi + 1) & 3;
        }
      }
    }
    if (l1 > l0) harm_index = (harm_index + (l1 - l0)) & 3;
    cx.sync();
  }
  if (pst) { /* esbr_envcal.c:861-864, :873-899: what a PVC frame behind this one reads */
    XS_PAR(k, 0, 64) {
      pst->harm_flag_varlen_prev[k] = w->harm_prev[k];
      pst->harm_flag_varlen[k] = w->harmonics[k];
    }
    XS_PAR(k, 0, num_nf) pst->prev_noise_level[k] = sd->flt_noise_floor[(f->num_noise_env - 1) * num_nf + k];
    XS_ONE {
      pst->prev_freq_res[0] = f->freq_res[0];
      pst->prev_freq_res[1] = f->freq_res[1];
      if (num_env == 1) pst->var_len_id_prev = 0;
      else if (num_env == 2) pst->var_len_id_prev = 1;
      pst->esbr_start_up_pvc = start_up_pvc;
    }
  }
  XS_PAR(k, 0, 64) if (k >= sb_start) st->harm_flag_prev[k] = w->harmonics[k - sb_start];
  XS_ONE {
    st->env_short_flag_prev = trans_env == num_env ? 0 : -1;
    st->harm_index = harm_index;
    st->phase_index = phase_index;
    st->esbr_start_up = start_up;
  }
  cx.sync();
  return 0;
}

#endif
