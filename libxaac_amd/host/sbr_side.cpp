/*
 * sbr_side.cpp -- SBR / PS side-info decoder on the host (see sbr_side.h for the reference map).
 * Tables: tables_sbr_side.inc (code books, FIXFIX grids, log2), ../csrc/tables_sbr.inc (reciprocal table of
 * ixheaacd_fix_mant_div), both generated from the compiled reference's ROM.
 */
#include "sbr_side.h"

#include <math.h>
#include <string.h>

#include "../csrc/fx.h"
#include "../csrc/tables_sbr.inc"
#include "tables_sbr_side.inc"

namespace {

enum { LO = 0, HI = 1 };
enum { DIR_FREQ = 0, DIR_TIME = 1 };
enum { FIXFIX = 0, FIXVAR = 1, VARFIX = 2, VARVAR = 3 };
const int kTimeSlots = 16, kOvSlots = 3;
const int kExpBits = 6, kMaskExp = 63, kMaskM = 0xffc0, kRounding = 32, kNrgExpOffset = 16, kNoiseExpOffset = 38;

/* ---- code books ----------------------------------------------------------------------------------------------------- */
struct Book {
  const uint32_t *code;
  const uint8_t *len;
  const int16_t *val;
  int n;
  uint16_t lut[1024]; /* the first ten bits -> entry (code words of up to ten bits), or 0xffff: built by build_luts() */
};
#define XS_BOOK(name) {xh_##name##_code, xh_##name##_len, xh_##name##_val, (int)sizeof(xh_##name##_len), {0}}
Book k_env_t_15 = XS_BOOK(sbr_env_t_15), k_env_f_15 = XS_BOOK(sbr_env_f_15), k_env_t_30 = XS_BOOK(sbr_env_t_30),
     k_env_f_30 = XS_BOOK(sbr_env_f_30), k_bal_t_15 = XS_BOOK(sbr_bal_t_15), k_bal_f_15 = XS_BOOK(sbr_bal_f_15),
     k_bal_t_30 = XS_BOOK(sbr_bal_t_30), k_bal_f_30 = XS_BOOK(sbr_bal_f_30), k_noise_t_30 = XS_BOOK(sbr_noise_t_30),
     k_noise_bal_t_30 = XS_BOOK(sbr_noise_bal_t_30);
Book k_ps[6] = {XS_BOOK(ps_iid_df), XS_BOOK(ps_iid_dt), XS_BOOK(ps_iid_df_fine),
                XS_BOOK(ps_iid_dt_fine), XS_BOOK(ps_icc_df), XS_BOOK(ps_icc_dt)};

void build_luts() { /* once, from xs_init (also when the first callers are parser threads: a function-local static) */
  Book *all[16] = {&k_env_t_15, &k_env_f_15, &k_env_t_30, &k_env_f_30, &k_bal_t_15, &k_bal_f_15, &k_bal_t_30, &k_bal_f_30,
                   &k_noise_t_30, &k_noise_bal_t_30, &k_ps[0], &k_ps[1], &k_ps[2], &k_ps[3], &k_ps[4], &k_ps[5]};
  for (Book *k : all) {
    for (int i = 0; i < 1024; i++) k->lut[i] = 0xffff;
    for (int e = 0; e < k->n; e++)
      if (k->len[e] <= 10) {
        const uint32_t first = k->code[e] >> 22, count = 1u << (10 - k->len[e]);
        for (uint32_t j = 0; j < count; j++) k->lut[first + j] = (uint16_t)e;
      }
  }
}

inline int huff(const Book &k, XhBits *br) {
  const uint32_t w = br->peek32();
  int lo = k.lut[w >> 22];
  if (lo == 0xffff) { /* a longer code word.  They are sorted: the last one not above the window is the prefix */
    lo = 0;
    int hi = k.n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (k.code[mid] <= w) lo = mid;
      else hi = mid - 1;
    }
  }
  br->skip(k.len[lo]);
  return k.val[lo];
}

/* ---- small arithmetic of the reference ---------------------------------------------------------------------------- */
int int_div(int num, int den) { /* freq_sca.c:60 */
  if (den == 0) return 0;
  int result = 0;
  while (den <= num) {
    int t = 0;
    while (num >= (den << (t + 1))) t++;
    result += 1 << t;
    num -= den * (1 << t);
  }
  return result;
}

void shellsort(int16_t *in, int n) { /* freq_sca.c:78 */
  int inc = 1;
  do inc = 3 * inc + 1;
  while (inc <= n);
  do {
    inc = int_div(inc, 3);
    for (int i = inc; i < n; i++) {
      const int v = in[i];
      int j = i, w;
      while ((w = in[j - inc]) > v) {
        in[j] = (int16_t)w;
        j -= inc;
        if (j < inc) break;
      }
      in[j] = (int16_t)v;
    }
  } while (inc > 1);
}

inline int16_t add16(int a, int b) { return (int16_t)(a + b); }

void mant_exp_add(int16_t m1, int16_t e1, int16_t m2, int16_t e2, int16_t *rm, int16_t *re) { /* basic_funcs.c:35 */
  int32_t a = m1, b = m2, e = e1 - e2;
  if (e < 0) {
    if (e < -31) e = -31;
    a >>= -e;
    e = e2;
  } else {
    if (e > 31) e = 31;
    b >>= e;
    e = e1;
  }
  int32_t m = a + b;
  if ((m < 0 ? -m : m) >= 0x8000) {
    m >>= 1;
    e++;
  }
  *rm = (int16_t)m;
  *re = (int16_t)e;
}

int mant_div(int16_t op1, int16_t op2, int16_t *res) { /* basic_funcs.c:66 */
  const int pre = fx_norm32(op2) - 16;
  int index = (fx_shlw(op2, pre) >> (16 - 3 - 8)) & 511;
  int post;
  if (index == 0) {
    post = fx_norm32(op1) - 16;
    *res = (int16_t)fx_shlw(op1, post);
  } else {
    index = (index - 1) >> 1;
    const int32_t ratio = (int32_t)xaac_sbr_inv_table[index] * (int32_t)op1;
    post = fx_norm32(ratio) - 1;
    *res = (int16_t)(fx_shlw(ratio, post) >> 15);
  }
  return pre - post;
}

/* ---- frequency band tables: freq_sca.c ----------------------------------------------------------------------------- */
int start_band(int fs_mapped, int start_freq) { /* :108, upsampling factor 2 */
  const int k0_min = (int)(((float)((fs_mapped < 32000 ? 3000 : (fs_mapped < 64000 ? 4000 : 5000)) * 2 * 64) / fs_mapped) + 0.5);
  static const int8_t off_16[16] = {-8, -7, -6, -5, -4, -3, -2, -1, 0, 1, 2, 3, 4, 5, 6, 7};
  static const int8_t off_22[16] = {-5, -4, -3, -2, -1, 0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 13};
  static const int8_t off_24[16] = {-5, -3, -2, -1, 0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 13, 16};
  static const int8_t off_32[16] = {-6, -4, -2, -1, 0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 13, 16};
  static const int8_t off_40[16] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 13, 15, 17, 19};
  static const int8_t off_48[16] = {-4, -2, -1, 0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 13, 16, 20};
  static const int8_t off_96[16] = {-2, -1, 0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 13, 16, 20, 24};
  static const int8_t off_x[16] = {0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 13, 16, 20, 24, 28, 33};
  const int8_t *o = off_x;
  switch (fs_mapped) {
    case 16000: o = off_16; break;
    case 22050: o = off_22; break;
    case 24000: o = off_24; break;
    case 32000: o = off_32; break;
    case 40000: o = off_40; break;
    case 44100: case 48000: case 64000: o = off_48; break;
    case 88200: case 96000: o = off_96; break;
  }
  return k0_min + o[start_freq];
}

int stop_band(int fs, int stop_freq) { /* :183 */
  const int k1_min = (int)(((float)((fs < 32000 ? 6000 : (fs < 64000 ? 8000 : 10000)) * 2 * 64) / fs) + 0.5);
  int16_t stop[14], diff[13];
  for (int i = 0; i <= 13; i++) stop[i] = (int16_t)(int32_t)(k1_min * pow(64.0 / k1_min, i / 13.0) + 0.5);
  for (int i = 0; i <= 12; i++) diff[i] = (int16_t)(stop[i + 1] - stop[i]);
  shellsort(diff, 13);
  int32_t r = k1_min;
  for (int i = 0; i < stop_freq; i++) r = fx_add_sat(r, diff[i]);
  return r;
}

int16_t freq_ratio(int16_t k_start, int16_t k_stop, int num_bands) { /* :504 */
  int32_t bandfactor = 0x3f000000, step = 0x20000000;
  int direction = 1;
  const int32_t start = fx_shl((int32_t)k_start, 24), stop = fx_shl((int32_t)k_stop, 24);
  int i = 0;
  do {
    i++;
    int32_t t = stop;
    for (int j = 0; j < num_bands; j++) t = fx_shlw((int32_t)(int16_t)(t >> 16) * (int32_t)(int16_t)(bandfactor >> 16), 1);
    if (t < start) {
      if (direction == 0) step = fx_shr(step, 1);
      direction = 1;
      bandfactor = fx_add_sat(bandfactor, step);
    } else {
      if (direction == 1) step = fx_shr(step, 1);
      direction = 0;
      bandfactor = fx_sub_sat(bandfactor, step);
    }
    if (i > 100) step = 0;
  } while (step > 0);
  return (int16_t)(bandfactor >> 16);
}

void calc_bands(int16_t *diff, int16_t start, int16_t stop, int num_bands) { /* :548 */
  const int16_t bandfactor = freq_ratio(start, stop, num_bands);
  int32_t previous = stop;
  int32_t exact = fx_shl_sat((int32_t)stop, 24);
  for (int i = num_bands - 1; i >= 0; i--) {
    exact = (int32_t)(int16_t)(exact >> 16) * (int32_t)bandfactor;
    const int32_t t = fx_add_sat(exact, 0x00400000);
    exact = fx_shlw(exact, 1);
    const int32_t current = (int16_t)fx_shr(t, 23);
    diff[i] = (int16_t)(previous - current);
    previous = current;
  }
}

int master_table(XsHeader *h) { /* :232-502 (upsampling factor 2) */
  const int fs = h->out_sampling_freq;
  int fs_mapped;
  if (fs < 18783) fs_mapped = 16000;
  else if (fs < 23004) fs_mapped = 22050;
  else if (fs < 27713) fs_mapped = 24000;
  else if (fs < 35777) fs_mapped = 32000;
  else if (fs < 42000) fs_mapped = 40000;
  else if (fs < 46009) fs_mapped = 44100;
  else if (fs < 55426) fs_mapped = 48000;
  else if (fs < 75132) fs_mapped = 64000;
  else if (fs < 92017) fs_mapped = 88200;
  else fs_mapped = 96000;
  int16_t k0 = (int16_t)start_band(fs_mapped, h->start_freq), k2;
  if (h->stop_freq < 14) k2 = (int16_t)stop_band(fs, h->stop_freq);
  else if (h->stop_freq == 14) k2 = (int16_t)(2 * k0);
  else k2 = (int16_t)(3 * k0);
  if (k2 > 64) k2 = 64;
  if (k2 - k0 > 48 || k2 <= k0) return -1;
  if (fs == 44100 && k2 - k0 > 35) return -1;
  if (fs >= 48000 && k2 - k0 > 32) return -1;
  int16_t vec_dk[50 + 50];
  int16_t *f = h->f_master;
  int num_mf;
  if (h->freq_scale == 0) {
    int dk, num_bands;
    if (h->alter_scale == 0) {
      dk = 1;
      num_bands = (int16_t)(k2 - k0);
      num_bands -= num_bands & 1;
    } else {
      dk = 2;
      num_bands = ((int16_t)((k2 - k0) + 2) >> 2) << 1;
    }
    if (num_bands < 1) return -1;
    const int achieved = k0 + (num_bands << (dk - 1));
    int k2_diff = k2 - achieved, incr = 0, k = 0;
    for (int i = 0; i < num_bands; i++) vec_dk[i] = (int16_t)dk;
    if (k2_diff < 0) incr = 1, k = 0;
    if (k2_diff > 0) incr = -1, k = num_bands - 1;
    while (k2_diff != 0) {
      vec_dk[k] = (int16_t)(vec_dk[k] - incr);
      k = (int16_t)(k + incr);
      k2_diff += incr;
    }
    f[0] = k0;
    for (int i = 1; i <= num_bands; i++) f[i] = (int16_t)(f[i - 1] + vec_dk[i - 1]);
    num_mf = num_bands;
  } else {
    const int bands = h->freq_scale == 1 ? 12 : (h->freq_scale == 2 ? 10 : 8);
    int16_t *dk0 = vec_dk, *dk1 = vec_dk + 50;
    if (10000 * k2 > 22449 * k0) {
      const int16_t k1 = (int16_t)(k0 << 1);
      const int nb0 = bands;
      int32_t nb1 = bands * (xh_log_dual_is[k2] - xh_log_dual_is[k1]);
      if (h->alter_scale) nb1 = (int32_t)(((int64_t)nb1 * 0x6276) >> 15);
      nb1 = ((nb1 + 0x1000) >> 13) << 1;
      if (nb0 < 1 || nb1 < 1) return -1;
      if (nb1 > 50 || nb0 + nb1 > XAAC_SBR_MAX_FREQ_COEFFS) return -1; /* beyond the reference's own arrays (vec_dk, f_master_tbl) */
      calc_bands(dk0, k0, k1, nb0);
      shellsort(dk0, nb0);
      f[0] = k0;
      for (int i = 1; i <= nb0; i++) f[i] = (int16_t)(f[i - 1] + dk0[i - 1]);
      calc_bands(dk1, k1, k2, nb1);
      shellsort(dk1, nb1);
      if (dk1[0] < dk0[nb0 - 1]) {
        int16_t change = (int16_t)(dk0[nb0 - 1] - dk1[0]);
        const int16_t half = (int16_t)((int16_t)(dk1[nb1 - 1] - dk1[0]) >> 1);
        if (change > half) change = half;
        dk1[0] = (int16_t)(dk1[0] + change);
        dk1[nb1 - 1] = (int16_t)(dk1[nb1 - 1] - change);
        shellsort(dk1, nb1);
      }
      f[nb0] = k1;
      for (int i = 1; i <= nb1; i++) f[nb0 + i] = (int16_t)(f[nb0 + i - 1] + dk1[i - 1]);
      num_mf = nb0 + nb1;
    } else {
      int32_t nb0 = bands * (xh_log_dual_is[k2] - xh_log_dual_is[k0]);
      nb0 = ((nb0 + 0x1000) >> 13) << 1;
      if (nb0 < 1) return -1;
      if (nb0 > 50) return -1;
      calc_bands(dk0, k0, k2, nb0);
      shellsort(dk0, nb0);
      if (dk0[0] == 0) return -1;
      f[0] = k0;
      for (int i = 1; i <= nb0; i++) f[i] = (int16_t)(f[i - 1] + dk0[i - 1]);
      num_mf = nb0;
    }
  }
  if (num_mf < 1) return -1;
  h->num_mf_bands = (int16_t)num_mf;
  return 0;
}

int calc_freq_tables(XsHeader *h) { /* :668-712 with :572 and :613 */
  if (master_table(h) || h->xover_band > h->num_mf_bands) return -1;
  { /* high and low resolution tables */
    const int16_t *m = h->f_master + h->xover_band;
    const int num_hf = h->num_mf_bands - h->xover_band;
    int16_t *lo = h->tbl_lo, *hi = h->tbl_hi;
    int k = 0;
    *lo++ = *hi++ = *m++;
    k++;
    if (num_hf & 1) {
      *lo++ = *hi++ = *m++;
      k++;
    }
    for (; k <= num_hf; k++) {
      *hi++ = *m++;
      k++;
      *lo++ = *hi++ = *m++;
    }
    h->num_sf_bands[LO] = (int16_t)((num_hf + 1) >> 1);
    h->num_sf_bands[HI] = (int16_t)num_hf;
  }
  const int num_lf = h->num_sf_bands[LO];
  if (num_lf <= 0 || num_lf > (XAAC_SBR_MAX_FREQ_COEFFS >> 1)) return -1;
  const int16_t lsb = h->tbl_lo[0], usb = h->tbl_lo[num_lf];
  h->sub_band_start = lsb;
  if (lsb > 32 || lsb >= usb) return -1;
  { /* noise floor table */
    const int16_t k2 = h->tbl_hi[h->num_sf_bands[HI]], kx = h->tbl_hi[0];
    int32_t t;
    if (h->noise_bands == 0) {
      t = 1;
    } else {
      t = (xh_log_dual_is[k2] - xh_log_dual_is[kx]) * h->noise_bands;
      t = (t + 0x800) >> 12;
      if (t == 0) t = 1;
    }
    if (t > XAAC_SBR_MAX_NOISE_COEFFS) return -1;
    h->num_nf_bands = h->num_if_bands = (int16_t)t;
    int16_t num = (int16_t)num_lf, den = (int16_t)t, i_k = 0;
    h->tbl_noise[0] = h->tbl_lo[0];
    for (int k = 1; k <= t; k++) {
      i_k = (int16_t)(i_k + int_div(num, den));
      h->tbl_noise[k] = h->tbl_lo[i_k];
      num = (int16_t)(num_lf - i_k);
      den = (int16_t)(den - 1);
    }
  }
  h->sub_band_start = lsb;
  h->sub_band_end = usb;
  return 0;
}

/* ---- patches and limiter bands: sbrdec_lpfuncs.c ----------------------------------------------------------------------- */
int16_t closest_entry(int goal, const int16_t *f, int num, int up) { /* :232 */
  if (goal <= f[0]) return f[0];
  if (goal >= f[num]) return f[num];
  int i;
  if (up) {
    i = 0;
    while (f[i] < goal) i++;
  } else {
    i = num;
    while (f[i] > goal) i--;
  }
  return f[i];
}

int reset_hf_generator(XsHeader *h) { /* :250-440 (AAC-LC object types) */
  const int16_t *f = h->f_master;
  const int num_mf = h->num_mf_bands, usb = h->sub_band_end, lsb = f[0];
  const int16_t xover_offset = (int16_t)(h->sub_band_start - lsb);
  int goal;
  if (lsb < 1 + 4) return 1;
  switch (h->out_sampling_freq) {
    case 16000: case 22050: case 24000: case 32000: goal = 64; break;
    case 44100: goal = 46; break;
    case 48000: goal = 43; break;
    case 64000: goal = 32; break;
    case 88200: goal = 23; break;
    case 96000: goal = 21; break;
    default: return 0;
  }
  goal = closest_entry(goal, f, num_mf, 1);
  { /* abs16_sat */
    int d = (int16_t)(goal - usb);
    if (d < 0) d = -d;
    if (d < 4) goal = usb;
  }
  int src_start = 1 + xover_offset, sb = lsb + xover_offset, patch = 0, flag_break_1 = 0;
  if (goal < sb && lsb > src_start) return -1;
  while (sb - usb < 0 && patch < XAAC_SBR_MAX_PATCHES) {
    xaac_sbr_patch *p = &h->patch[patch];
    int flag_break = 0;
    p->guard_start_band = (int16_t)sb;
    p->dst_start_band = (int16_t)sb; /* GUARDBANDS = 0 */
    int n = goal - sb;
    if (n <= 0 && n - (lsb - src_start) < 0) flag_break = 1;
    if (n - (lsb - src_start) >= 0) {
      int stride = (int16_t)((sb - src_start) & ~1);
      n = lsb - (sb - stride);
      n = closest_entry(sb + n, f, num_mf, 0);
      n -= sb;
    }
    int stride = (int16_t)(((n + sb) - lsb + 1) & ~1);
    if (n > 0) {
      p->src_start_band = (int16_t)(sb - stride);
      p->dst_end_band = (int16_t)stride;
      p->num_bands_in_patch = (int16_t)n;
      p->src_end_band = (int16_t)(p->src_start_band + n);
      sb += p->num_bands_in_patch;
      patch++;
    }
    src_start = 1;
    int d = (int16_t)(sb - goal);
    if (d < 0) d = -d;
    const int abs_sb = d - 3;
    if (n <= 0 && flag_break_1 == 1) break;
    if (abs_sb < 0) goal = usb;
    else if (flag_break == 1) break;
    flag_break_1 = n <= 0;
  }
  patch--;
  if (patch > 0 && h->patch[patch].num_bands_in_patch < 3) {
    patch--;
    sb = h->patch[patch].dst_start_band + h->patch[patch].num_bands_in_patch;
  }
  if (patch >= XAAC_SBR_MAX_PATCHES) return -1;
  h->num_patches = (int16_t)(patch + 1);
  int hi = 0;
  for (patch = 0; patch < h->num_patches; patch++) {
    if (h->patch[patch].src_start_band < sb) sb = h->patch[patch].src_start_band;
    if (h->patch[patch].src_end_band > hi) hi = h->patch[patch].src_end_band;
  }
  if (sb > hi) return -2;
  h->start_patch = (int16_t)sb;
  h->stop_patch = (int16_t)hi;
  memcpy(h->bw_borders, &h->tbl_noise[1], sizeof(int16_t) * (size_t)h->num_nf_bands);
  return 0;
}

void derive_lim_bands(XsHeader *h) { /* :72-193 */
  const int num_low = h->num_sf_bands[LO], num_patches = h->num_patches;
  const int16_t lo0 = h->tbl_lo[0], end = h->tbl_lo[num_low];
  int nr_lim;
  if (h->limiter_bands == 0) {
    h->tbl_lim[0] = 0;
    h->tbl_lim[1] = (int16_t)(end - lo0);
    nr_lim = 1;
  } else {
    static const int16_t per_oct[4] = {0x2000, 0x2666, 0x4000, 0x6000};
    int16_t lim[XAAC_SBR_MAX_FREQ_COEFFS / 2 + XAAC_SBR_MAX_PATCHES + 1], borders[XAAC_SBR_MAX_PATCHES + 1];
    int k;
    for (k = 0; k < num_patches; k++) borders[k] = (int16_t)(h->patch[k].guard_start_band - lo0);
    borders[k] = (int16_t)(end - lo0);
    for (k = 0; k <= num_low; k++) lim[k] = (int16_t)(h->tbl_lo[k] - lo0);
    for (k = 1; k < num_patches; k++) lim[num_low + k] = borders[k];
    const int total = nr_lim = num_low + num_patches - 1;
    shellsort(lim, total + 1);
    k = 1;
    int k_1 = 0;
    const int16_t lim_bands = per_oct[h->limiter_bands];
    while (k - total <= 0) {
      const int k2 = lim[k] + lo0, kx = lim[k_1] + lo0;
      const int16_t oct = (int16_t)(xh_log_dual_is[k2] - xh_log_dual_is[kx]);
      const int16_t t = (int16_t)(((int32_t)lim_bands * (int32_t)oct) >> 15);
      if (t < 0x01f6) {
        if (lim[k_1] == lim[k]) {
          lim[k] = end;
          nr_lim--;
          k++;
          continue;
        }
        int at_k = 0, at_k_1 = 0;
        for (int i = 0; i <= num_patches; i++) {
          if (lim[k] == borders[i]) at_k = 1;
          if (lim[k_1] == borders[i]) at_k_1 = 1;
        }
        if (!at_k) {
          lim[k] = end;
          nr_lim--;
          k++;
          continue;
        }
        if (!at_k_1) {
          lim[k_1] = end;
          nr_lim--;
        }
      }
      k_1 = k;
      k++;
    }
    shellsort(lim, total + 1);
    memcpy(h->tbl_lim, lim, sizeof(int16_t) * (size_t)(nr_lim + 1));
  }
  h->num_lf_bands = (int16_t)nr_lim;
}

/* ---- payload parsing: env_extr.c ------------------------------------------------------------------------------------ */
#define SBR_RESET 1

int read_header(XsHeader *h, XhBits *br, int stereo) { /* :354-511 (not USAC) */
  const XsHeader prev = *h;
  uint32_t t = br->get(12);
  h->amp_res = (int)((t & 0x800) >> 11);
  h->start_freq = (int)((t & 0x780) >> 7);
  h->stop_freq = (int)((t & 0x78) >> 3);
  h->xover_band = (int)(t & 7);
  t = br->get(4);
  const int extra_1 = (int)((t & 2) >> 1), extra_2 = (int)(t & 1);
  h->channel_mode = stereo ? XS_SBR_STEREO : XS_SBR_MONO;
  if (extra_1) {
    t = br->get(5);
    h->freq_scale = (int)((t & 0x18) >> 3);
    h->alter_scale = (int)((t & 4) >> 2);
    h->noise_bands = (int)(t & 3);
  } else {
    h->freq_scale = 2, h->alter_scale = 1, h->noise_bands = 2;
  }
  if (extra_2) {
    t = br->get(6);
    h->limiter_bands = (int)((t & 0x30) >> 4);
    h->limiter_gains = (int)((t & 0xc) >> 2);
    h->interpol_freq = (int)((t & 2) >> 1);
    h->smoothing_mode = (int)(t & 1);
  } else {
    h->limiter_bands = 2, h->limiter_gains = 2, h->interpol_freq = 1, h->smoothing_mode = 1;
  }
  if (h->sync_state != XS_ACTIVE || prev.start_freq != h->start_freq || prev.stop_freq != h->stop_freq ||
      prev.xover_band != h->xover_band || prev.freq_scale != h->freq_scale || prev.alter_scale != h->alter_scale ||
      prev.noise_bands != h->noise_bands)
    return SBR_RESET;
  return 0;
}

int read_grid(XhBits *br, XsFrameInfo *fi) { /* :1740-1959, 16 time slots */
  static const int pointer_bits[7] = {1, 2, 2, 3, 3, 3, 3};
  int num_env = 0;
  const int frame_class = (int)br->get(2);
  fi->frame_class = (int16_t)frame_class;
  switch (frame_class) {
    case FIXFIX: {
      const uint32_t t = br->get(3);
      const int e = (int)((t & 6) >> 1);
      const int16_t *row = xh_sbr_frame_info + 24 * e;
      fi->frame_class = row[0], fi->num_env = row[1], fi->transient_env = row[2], fi->num_noise_env = row[3];
      memcpy(fi->border_vec, row + 4, sizeof(fi->border_vec));
      memcpy(fi->freq_res, row + 13, sizeof(fi->freq_res));
      memcpy(fi->noise_border_vec, row + 21, sizeof(fi->noise_border_vec));
      num_env = 1 << e;
      if (!(t & 1)) memset(fi->freq_res, 0, sizeof(int16_t) * (size_t)num_env);
      break;
    }
    case FIXVAR: {
      uint32_t t = br->get(4);
      const int num_rel = (int)(t & 3);
      int border = (int)(t >> 2) + kTimeSlots;
      num_env = num_rel + 1;
      fi->border_vec[0] = 0;
      fi->border_vec[num_env] = (int16_t)border;
      for (int k = num_rel; k > 0; k--) {
        border -= ((int)br->get(2) << 1) + 2;
        if (border < 0) border = 0;
        fi->border_vec[k] = (int16_t)border;
      }
      const int pointer = (int)br->get(pointer_bits[num_rel]);
      if (pointer - (num_rel + 1) > 0) return 0;
      for (int k = num_rel; k >= 0; k--) fi->freq_res[k] = (int16_t)br->get1();
      fi->transient_env = pointer ? (int16_t)(num_env + 1 - pointer) : (int16_t)-1;
      fi->noise_border_vec[1] = pointer <= 1 ? fi->border_vec[num_rel] : fi->border_vec[fi->transient_env];
      break;
    }
    case VARFIX: {
      uint32_t t = br->get(4);
      const int num_rel = (int)(t & 3);
      int border = (int)(t >> 2), k;
      num_env = num_rel + 1;
      fi->border_vec[0] = (int16_t)border;
      for (k = 1; k <= num_rel; k++) {
        border += ((int)br->get(2) << 1) + 2;
        if (border > kTimeSlots) border = kTimeSlots;
        fi->border_vec[k] = (int16_t)border;
      }
      fi->border_vec[k] = kTimeSlots;
      const int pointer = (int)br->get(pointer_bits[num_rel]);
      if (pointer - (num_rel + 1) > 0) return 0;
      fi->transient_env = pointer <= 1 ? (int16_t)-1 : (int16_t)(pointer - 1);
      for (k = 0; k <= num_rel; k++) fi->freq_res[k] = (int16_t)br->get1();
      if (pointer == 0) fi->noise_border_vec[1] = fi->border_vec[1];
      else if (pointer == 1) fi->noise_border_vec[1] = fi->border_vec[num_rel];
      else fi->noise_border_vec[1] = fi->border_vec[fi->transient_env];
      break;
    }
    case VARVAR: {
      const uint32_t t = br->get(8);
      const int trail_abs = (int)((t & 0x30) >> 4) + kTimeSlots, rel_trail = (int)((t & 0xc) >> 2), rel_lead = (int)(t & 3);
      const int lead_abs = (int)(t >> 6);
      num_env = rel_trail + rel_lead + 1;
      int border = lead_abs, k;
      fi->border_vec[0] = (int16_t)border;
      for (k = 1; k <= rel_trail; k++) {
        border += ((int)br->get(2) << 1) + 2;
        fi->border_vec[k] = (int16_t)border;
      }
      border = trail_abs;
      int i = num_env;
      fi->border_vec[i] = (int16_t)border;
      for (k = 0; k < rel_lead; k++) {
        border -= ((int)br->get(2) << 1) + 2;
        fi->border_vec[--i] = (int16_t)border;
      }
      const int pointer = (int)br->get(pointer_bits[rel_trail + rel_lead]);
      if (pointer - (rel_trail + rel_lead + 1) > 0) return 0;
      fi->transient_env = pointer ? (int16_t)(num_env + 1 - pointer) : (int16_t)-1;
      for (k = 0; k < num_env; k++) fi->freq_res[k] = (int16_t)br->get1();
      fi->noise_border_vec[0] = (int16_t)lead_abs;
      if (num_env == 1) {
        fi->noise_border_vec[1] = (int16_t)trail_abs;
      } else {
        fi->noise_border_vec[1] = pointer <= 1 ? fi->border_vec[num_env - 1] : fi->border_vec[fi->transient_env];
        fi->noise_border_vec[2] = (int16_t)trail_abs;
      }
      break;
    }
  }
  fi->num_env = (int16_t)num_env;
  fi->num_noise_env = num_env == 1 ? 1 : 2;
  if (frame_class == VARFIX || frame_class == FIXVAR) {
    fi->noise_border_vec[0] = fi->border_vec[0];
    fi->noise_border_vec[fi->num_noise_env] = fi->border_vec[num_env];
  }
  return 1;
}

int validate_grid(const XsFrameInfo *fi) { /* :530-593, 16 time slots */
  const int n = fi->num_env, nn = fi->num_noise_env;
  if (n < 1 || n > XAAC_SBR_MAX_ENVELOPES) return 0;
  if (nn > XAAC_SBR_MAX_NOISE_ENVELOPES) return 0;
  const int start = fi->border_vec[0], end = fi->border_vec[n];
  if (fi->transient_env > n) return 0;
  if (start < 0 || start >= end) return 0;
  if (start > kOvSlots) return 0;
  if (end < kTimeSlots) return 0;
  if (end > kTimeSlots + kOvSlots) return 0;
  for (int i = 0; i < n; i++)
    if (fi->border_vec[i] > fi->border_vec[i + 1]) return 0;
  if (n == 1 && nn > 1) return 0;
  if (start != fi->noise_border_vec[0] || end != fi->noise_border_vec[nn]) return 0;
  for (int i = 0; i < nn; i++) {
    int j;
    for (j = 0; j < n; j++)
      if (fi->border_vec[j] == fi->noise_border_vec[i]) break;
    if (j == n) return 0;
  }
  return 1;
}

void read_dtdf(XsFrameData *f, XhBits *br) { /* :1231-1277 */
  for (int i = 0; i < f->fi.num_env; i++) f->dir_env[i] = (int16_t)br->get1();
  for (int i = 0; i < f->fi.num_noise_env; i++) f->dir_noise[i] = (int16_t)br->get1();
}

/* :1279-1359: the start value of an envelope coded along frequency, then code words (delta = value - lav) */
void read_deltas(XsFrameData *f, XhBits *br, const Book &hcb_t, const Book &hcb_f, const int16_t *no_band, int num_env, int comp,
                 int start_bits, int start_bits_bal, int is_noise, int lav) {
  const int16_t *dir = is_noise ? f->dir_noise : f->dir_env;
  int16_t *sf = is_noise ? f->noise_floor : f->env_sf;
  float *sff = is_noise ? f->flt_noise_floor : f->flt_env_sf;
  const int bal = f->coupling_mode == XS_COUPLING_BAL;
  const int bits = bal ? start_bits_bal : start_bits, shift = bal ? comp : 0;
  int offset = 0;
  for (int j = 0; j < num_env; j++) {
    const int d = dir[j];
    if (d == DIR_FREQ) {
      sf[offset] = (int16_t)(br->get(bits) << shift);
      sff[offset] = sf[offset];
    }
    const Book &h = d == DIR_FREQ ? hcb_f : hcb_t;
    for (int i = 1 - d; i < no_band[j]; i++) {
      sf[offset + i] = (int16_t)((huff(h, br) - lav) * (1 << comp));
      sff[offset + i] = sf[offset + i];
    }
    offset += no_band[j];
  }
}

int read_envelopes(const XsHeader *h, XsFrameData *f, XhBits *br) { /* :1417-1513 */
  int amp_res = h->amp_res;
  const int num_env = f->fi.num_env;
  if (f->fi.frame_class == FIXFIX && num_env == 1) amp_res = 0;
  f->amp_res = (int16_t)amp_res;
  int16_t no_band[XAAC_SBR_MAX_ENVELOPES];
  f->num_env_sfac = 0;
  for (int i = 0; i < num_env; i++) {
    no_band[i] = h->num_sf_bands[f->fi.freq_res[i]];
    f->num_env_sfac = (int16_t)(f->num_env_sfac + no_band[i]);
  }
  if (f->num_env_sfac > XAAC_SBR_MAX_ENV_VALUES) return 0;
  const int bal = f->coupling_mode == XS_COUPLING_BAL;
  const Book &t = bal ? (amp_res ? k_bal_t_30 : k_bal_t_15) : (amp_res ? k_env_t_30 : k_env_t_15);
  const Book &fq = bal ? (amp_res ? k_bal_f_30 : k_bal_f_15) : (amp_res ? k_env_f_30 : k_env_f_15);
  read_deltas(f, br, t, fq, no_band, num_env, bal ? 1 : 0, amp_res ? 6 : 7, amp_res ? 5 : 6, 0,
              bal ? (amp_res ? 12 : 24) : (amp_res ? 31 : 60));
  return 1;
}

void read_noise(const XsHeader *h, XsFrameData *f, XhBits *br) { /* :1361-1415 */
  int16_t no_band[XAAC_SBR_MAX_NOISE_ENVELOPES];
  for (int i = 0; i < f->fi.num_noise_env; i++) no_band[i] = h->num_nf_bands;
  const int bal = f->coupling_mode == XS_COUPLING_BAL;
  read_deltas(f, br, bal ? k_noise_bal_t_30 : k_noise_t_30, bal ? k_bal_f_30 : k_env_f_30, no_band, f->fi.num_noise_env,
              bal ? 1 : 0, 5, 5, 1, bal ? 12 : 31);
}

void read_sines(const XsHeader *h, XsFrameData *f, XhBits *br) {
  if (br->get1()) {
    for (int i = 0; i < h->num_sf_bands[HI]; i++) f->add_harmonics[i] = (uint8_t)br->get1();
  } else {
    memset(f->add_harmonics, 0, sizeof(f->add_harmonics));
  }
}

/* PS payload: sbrdec_lpfuncs.c:587-735; returns the bits it took, or -1 (IA_FATAL_ERROR) */
int read_ps(XsPs *ps, XhBits *br, int bits_left) {
  static const int16_t num_env_tab[4] = {0, 1, 2, 4};
  static const int num_bands[3] = {10, 20, 34};
  const size_t at = br->pos;
  if (br->get1()) { /* enable_ps_header */
    ps->enable_iid = br->get1();
    if (ps->enable_iid) ps->iid_mode = (int)br->get(3);
    if (ps->iid_mode > 2) {
      ps->iid_quant = 1;
      ps->iid_mode -= 3;
    } else {
      ps->iid_quant = 0;
    }
    ps->enable_icc = br->get1();
    if (ps->enable_icc) ps->icc_mode = (int)br->get(3);
    ps->enable_ext = br->get1();
    if (ps->icc_mode > 2) ps->icc_mode -= 3;
    ps->freq_res_ipd = ps->iid_mode;
    if (ps->freq_res_ipd > 2) return -1;
  }
  if ((ps->enable_iid && ps->iid_mode > 2) || (ps->enable_icc && ps->icc_mode > 2)) {
    ps->data_present = 0;
    bits_left -= (int)(br->pos - at);
    while (bits_left > 8) {
      br->get(8);
      bits_left -= 8;
    }
    if (bits_left >= 0) br->get(bits_left);
    return (int)(br->pos - at);
  }
  ps->frame_class = br->get1();
  const int t = (int)br->get(2);
  if (ps->frame_class == 0) {
    ps->num_env = num_env_tab[t];
  } else {
    ps->num_env = 1 + t;
    for (int e = 1; e < ps->num_env + 1; e++) ps->border_position[e] = (int16_t)(br->get(5) + 1);
  }
  if (ps->enable_iid) {
    const Book &df = k_ps[ps->iid_quant ? 2 : 0], &dt = k_ps[ps->iid_quant ? 3 : 1];
    for (int e = 0; e < ps->num_env; e++) {
      ps->iid_dt[e] = (uint8_t)br->get1();
      for (int b = 0; b < num_bands[ps->iid_mode]; b++) ps->iid_par[e][b] = (int16_t)huff(ps->iid_dt[e] ? dt : df, br);
    }
  }
  if (ps->enable_icc) {
    for (int e = 0; e < ps->num_env; e++) {
      ps->icc_dt[e] = (uint8_t)br->get1();
      for (int b = 0; b < num_bands[ps->icc_mode]; b++) ps->icc_par[e][b] = (int16_t)huff(k_ps[ps->icc_dt[e] ? 5 : 4], br);
    }
  }
  if (ps->enable_ext) {
    int cnt = br->left() < 4 ? (int)br->get((int)br->left()) : (int)br->get(4);
    if (cnt == 15) cnt += (int)br->get(8);
    while (cnt--) br->get(8);
  }
  ps->data_present = 1;
  return (int)(br->pos - at);
}

/* :595-714: the ENHSBR extension of one element: patching mode, over-sampling flag, pitch; returns the bits it took */
int read_enh(XhBits *br, XsFrameData *f, int cpe, int *pre_flatten) {
  int bits = 1;
  *pre_flatten = br->get1(); /* header pre_proc_flag: stays until the next ENHSBR element (:602) */
  auto one = [&](XsFrameData *a, XsFrameData *b) {
    const int mode = br->get1();
    bits += 1;
    int over = 0, pitch = 0;
    if (mode == 0) {
      over = br->get1();
      bits += 2;
      if (br->get1()) {
        pitch = (int)br->get(7);
        bits += 7;
      }
    }
    a->patching_mode = mode, a->over_sampling = over, a->pitch_in_bins = pitch;
    if (b) b->patching_mode = mode, b->over_sampling = over, b->pitch_in_bins = pitch;
  };
  if (!cpe) {
    one(&f[0], nullptr);
  } else if (f[0].coupling_mode) {
    one(&f[0], &f[1]);
  } else {
    one(&f[0], nullptr);
    one(&f[1], nullptr);
  }
  if (bits < 6) {
    br->get(6 - bits);
    bits = 6;
  }
  return bits;
}

/* :716-795: extended data (PS, ENHSBR); returns -1 for the fatal error, 0 otherwise.  ps == NULL: the reference stops
   reading at a PS element.  enh: the eSBR interpretation (the ENHSBR element is read; otherwise it is skipped) */
int read_extension(XsHeader *h, XsPs *ps, XhBits *br, XsFrameData *f = nullptr, int cpe = 0, int enh = 0) {
  if (!br->get1()) return 0;
  int cnt = (int)br->get(4);
  if (cnt == 15) cnt += (int)br->get(8);
  int left = cnt << 3, ps_read = 0;
  while (left > 7) {
    int id = (int)br->get(2);
    if (id == 3 && !enh) id = -1; /* EXTENSION_ID_ENHSBR_CODING without the eSBR tools */
    left -= 2;
    if (id == 3) {
      left -= read_enh(br, f, cpe, &h->pre_flatten);
      continue;
    }
    if (id == 2) { /* EXTENSION_ID_PS_CODING */
      if (!ps) return 0;
      if (!ps_read) {
        const int used = read_ps(ps, br, left);
        if (used < 0) return -1;
        left -= used;
        if (left < 0) return 0;
        h->channel_mode = XS_PS_STEREO;
        ps_read = 1;
        continue;
      }
      /* a second PS element in one frame falls through to the enhanced-SBR reader in the reference: not built */
      id = -1;
    }
    const int bytes = left >> 3;
    br->skip(8 * (size_t)bytes);
    left -= bytes << 3;
  }
  if (left < 0) return 0;
  br->get(left);
  return 0;
}

/* :860-975; returns frame_status (1 ok, 0 bad) or -1 (fatal) */
int read_sce(XsHeader *h, XsFrameData *f, XsPs *ps, XhBits *br, int enh) {
  f->coupling_mode = XS_COUPLING_OFF;
  if (br->get1()) br->get(4);
  if (!read_grid(br, &f->fi)) return 0;
  if (!validate_grid(&f->fi)) return 0;
  read_dtdf(f, br);
  if (f->dir_env[0] == DIR_FREQ) h->err_flag = 0;
  for (int i = 0; i < h->num_if_bands; i++) {
    f->invf_mode_prev[i] = f->invf_mode[i];
    f->invf_mode[i] = (int32_t)br->get(2);
  }
  if (!read_envelopes(h, f, br)) return 0;
  read_noise(h, f, br);
  read_sines(h, f, br);
  if (read_extension(h, ps, br, f, 0, enh) < 0) return -1;
  return 1;
}

/* :977-1229 */
int read_cpe(XsHeader *h, XsFrameData *f, XhBits *br, int enh) {
  if (br->get1()) br->get(8);
  if (h->channel_mode != XS_SBR_STEREO) {
    h->sync_state = XS_UPSAMPLING;
    return 0;
  }
  const int coupling = br->get1();
  f[0].coupling_mode = coupling ? XS_COUPLING_LEVEL : XS_COUPLING_OFF;
  f[1].coupling_mode = coupling ? XS_COUPLING_BAL : XS_COUPLING_OFF;
  int num_ch = 2;
  for (int i = 0; i < num_ch; i++) {
    if (!read_grid(br, &f[i].fi)) return 0;
    if (!validate_grid(&f[i].fi)) return 0;
    if (coupling) {
      f[1].fi = f[0].fi;
      num_ch = 1;
    }
  }
  read_dtdf(&f[0], br);
  read_dtdf(&f[1], br);
  if (f[0].dir_env[0] == DIR_FREQ && f[1].dir_env[0] == DIR_FREQ) h->err_flag = 0;
  for (int k = 0; k < num_ch; k++)
    for (int i = 0; i < h->num_if_bands; i++) {
      f[k].invf_mode_prev[i] = f[k].invf_mode[i];
      f[k].invf_mode[i] = (int32_t)br->get(2);
    }
  if (coupling) {
    memcpy(f[1].invf_mode_prev, f[1].invf_mode, sizeof(int32_t) * (size_t)h->num_if_bands);
    memcpy(f[1].invf_mode, f[0].invf_mode, sizeof(int32_t) * (size_t)h->num_if_bands);
    if (!read_envelopes(h, &f[0], br)) return 0;
    read_noise(h, &f[0], br);
    if (!read_envelopes(h, &f[1], br)) return 0;
  } else {
    if (!read_envelopes(h, &f[0], br)) return 0;
    if (!read_envelopes(h, &f[1], br)) return 0;
    read_noise(h, &f[0], br);
  }
  read_noise(h, &f[1], br);
  read_sines(h, &f[0], br);
  read_sines(h, &f[1], br);
  if (read_extension(h, nullptr, br, f, 1, enh) < 0) return -1;
  return 1;
}

/* ---- delta decoding and dequantisation: env_dec.c ----------------------------------------------------------------------- */
void map_res_energy(int16_t v, int16_t *prev, int offset, int index, int res) { /* :88 */
  if (res == LO) {
    if (offset >= 0) {
      if (index < offset) {
        prev[index] = v;
      } else {
        const int i2 = 2 * index - offset;
        prev[i2] = prev[i2 + 1] = v;
      }
    } else {
      offset = -offset;
      if (index < offset) {
        const int i3 = 3 * index;
        prev[i3] = prev[i3 + 1] = prev[i3 + 2] = v;
      } else {
        const int i2 = 2 * index + offset;
        prev[i2] = prev[i2 + 1] = v;
      }
    }
  } else {
    prev[index] = v;
  }
}

void delta_decode_env(const XsHeader *h, XsFrameData *f, XsPrevData *p) { /* :122-237 */
  int16_t *prev = p->sfb_nrg_prev, *sf = f->env_sf;
  int offset = 2 * h->num_sf_bands[LO] - h->num_sf_bands[HI];
  for (int i = 0; i < f->fi.num_env; i++) {
    const int res = f->fi.freq_res[i], n = h->num_sf_bands[res];
    if (f->dir_env[i] == DIR_FREQ) {
      map_res_energy(sf[0], prev, offset, 0, res);
      for (int b = 1; b < n; b++) {
        sf[b] = (int16_t)(sf[b] + sf[b - 1]);
        map_res_energy(sf[b], prev, offset, b, res);
      }
    } else if (res == LO) {
      if (offset < 0) {
        offset = -offset; /* stays negated for the envelopes behind this one, as in the reference (:161) */
        const int tar = offset < n ? offset : n;
        int b;
        for (b = 0; b < tar; b++) {
          const int i3 = 3 * b;
          const int16_t t = add16(sf[b], prev[i3]);
          prev[i3] = prev[i3 + 1] = prev[i3 + 2] = t;
          sf[b] = t;
        }
        for (; b < n; b++) {
          const int i2 = 2 * b + offset;
          const int16_t t = add16(sf[b], prev[i2]);
          prev[i2] = prev[i2 + 1] = t;
          sf[b] = t;
        }
      } else {
        const int tar = offset < n ? offset : n;
        int b;
        for (b = 0; b < tar; b++) {
          sf[b] = add16(sf[b], prev[b]);
          prev[b] = sf[b];
        }
        for (; b < n; b++) {
          const int i2 = b < offset ? b : 2 * b - offset;
          const int16_t t = add16(sf[b], prev[i2]);
          prev[i2] = prev[i2 + 1] = t;
          sf[b] = t;
        }
      }
    } else {
      for (int b = 0; b < n; b++) {
        sf[b] = add16(sf[b], prev[b]);
        prev[b] = sf[b];
      }
    }
    sf += n;
  }
}

void lean_concealment(const XsHeader *h, XsFrameData *f, const XsPrevData *p) { /* sbrdec_lpfuncs.c:196-230 */
  f->amp_res = (int16_t)p->amp_res;
  f->coupling_mode = p->coupling_mode;
  f->max_qmf_subband_aac = p->max_qmf_subband_aac;
  memcpy(f->invf_mode, p->invf_mode, sizeof(f->invf_mode));
  f->fi.num_env = 1;
  const int16_t start = (int16_t)(p->end_position - kTimeSlots);
  f->fi.border_vec[0] = f->fi.noise_border_vec[0] = start;
  f->fi.border_vec[1] = f->fi.noise_border_vec[1] = kTimeSlots;
  f->fi.freq_res[0] = 1;
  f->fi.transient_env = -1;
  f->fi.num_noise_env = 1;
  f->num_env_sfac = h->num_sf_bands[HI];
  f->dir_env[0] = DIR_TIME;
  int target = f->coupling_mode == XS_COUPLING_BAL ? 12 : 0, step = 1;
  if (h->amp_res == 0) target <<= 1, step <<= 1;
  for (int i = 0; i < f->num_env_sfac; i++) f->env_sf[i] = (int16_t)(p->sfb_nrg_prev[i] > target ? -step : step);
  f->dir_noise[0] = DIR_TIME;
  memset(f->noise_floor, 0, sizeof(f->noise_floor));
  memset(f->add_harmonics, 0, sizeof(f->add_harmonics));
}

int timing_compensate(const XsHeader *h, XsFrameData *f, const XsPrevData *p) { /* env_dec.c:239-285 */
  XsFrameInfo *fi = &f->fi;
  int start_est = p->end_position - kTimeSlots;
  const int ref_len = fi->border_vec[1] - fi->border_vec[0];
  int new_len = fi->border_vec[1] - start_est;
  if (new_len <= 0) {
    new_len = ref_len;
    start_est = fi->border_vec[0];
  }
  int16_t delta = (int16_t)(xh_log_dual_is[ref_len] - xh_log_dual_is[new_len]);
  delta = (int16_t)(delta >> (16 - 0 - 3 - f->amp_res));
  fi->border_vec[0] = fi->noise_border_vec[0] = (int16_t)start_est;
  if (start_est < 0) return -1;
  if (f->coupling_mode != XS_COUPLING_BAL) {
    const int n = fi->freq_res[0] ? h->num_sf_bands[HI] : h->num_sf_bands[LO];
    for (int i = 0; i < n; i++) f->env_sf[i] = add16(f->env_sf[i], delta);
  }
  return 0;
}

int decode_envelope(XsHeader *h, XsFrameData *f, XsPrevData *p0, XsPrevData *p1, int enh) { /* :727-843 (no concealment option) */
  int t = p0->end_position - kTimeSlots;
  if (t < 0) return -1;
  t = f->fi.border_vec[0] - t;
  if (!h->err_flag_prev && !h->err_flag && t != 0) {
    if (f->dir_env[0] == DIR_TIME) h->err_flag = 1;
    else h->err_flag_prev = 1;
  }
  if (enh) { /* usac_flag | enh_sbr: no concealment, no timing compensation, no range check, no dequantisation (:759-842) */
    delta_decode_env(h, f, p0);
    for (int i = 0; i < f->num_env_sfac; i++) f->flt_env_sf[i] = (float)f->env_sf[i];
    return 0;
  }
  if (h->err_flag) {
    lean_concealment(h, f, p0);
    delta_decode_env(h, f, p0);
  } else {
    const int num = h->num_sf_bands[HI];
    if (h->err_flag_prev) {
      if (timing_compensate(h, f, p0)) return -1;
      if (f->coupling_mode != (int16_t)p0->coupling_mode) {
        if (p0->coupling_mode == XS_COUPLING_BAL) {
          memcpy(p0->sfb_nrg_prev, p1->sfb_nrg_prev, sizeof(int16_t) * (size_t)num);
        } else if (f->coupling_mode == XS_COUPLING_LEVEL) {
          for (int i = 0; i < num; i++) p0->sfb_nrg_prev[i] = (int16_t)(add16(p0->sfb_nrg_prev[i], p1->sfb_nrg_prev[i]) >> 1);
        } else if (f->coupling_mode == XS_COUPLING_BAL) {
          memset(p0->sfb_nrg_prev, 12, sizeof(int16_t) * (size_t)num); /* the reference's byte-wise memset (:801) */
        }
      }
    }
    int16_t saved[XAAC_SBR_MAX_FREQ_COEFFS];
    memcpy(saved, p0->sfb_nrg_prev, sizeof(saved));
    delta_decode_env(h, f, p0);
    { /* ixheaacd_check_env_data :287-320 */
      int bad = 0;
      const int16_t max_sf = (int16_t)(70 >> f->amp_res);
      for (int i = 0; i < f->num_env_sfac; i++) {
        if (f->env_sf[i] > max_sf) bad = 1;
        if (f->env_sf[i] < 0) f->env_sf[i] = 0;
      }
      for (int i = 0; i < num; i++) {
        if (p0->sfb_nrg_prev[i] < 0) p0->sfb_nrg_prev[i] = 0;
        else if (p0->sfb_nrg_prev[i] > max_sf) p0->sfb_nrg_prev[i] = max_sf;
      }
      if (bad) {
        h->err_flag = 1;
        memcpy(p0->sfb_nrg_prev, saved, sizeof(saved));
        return decode_envelope(h, f, p0, p1, enh);
      }
    }
  }
  { /* ixheaacd_dequant_env_data :322-343 */
    static const int32_t mant[2] = {0x4000, 0x5a80};
    const int a1 = 1 - f->amp_res;
    for (int i = 0; i < f->num_env_sfac; i++) {
      int e = f->env_sf[i];
      const int32_t m = mant[e & a1];
      e = (e >> a1) + 7 + kNrgExpOffset;
      f->env_sf[i] = (int16_t)(m | (e & kMaskExp));
    }
  }
  return 0;
}

int decode_noise(const XsHeader *h, XsFrameData *f, XsPrevData *p, int enh) { /* :396-494 */
  const int nb = h->num_nf_bands, ne = f->fi.num_noise_env;
  int16_t *nf = f->noise_floor;
  if (f->dir_noise[0] == DIR_FREQ) {
    for (int i = 1; i < nb; i++) nf[i] = (int16_t)(nf[i] + nf[i - 1]);
  } else {
    for (int i = 0; i < nb; i++) nf[i] = (int16_t)(nf[i] + p->prev_noise_level[i]);
  }
  if (ne > 1) {
    if (f->dir_noise[1] == DIR_FREQ) {
      for (int i = 1; i < nb; i++) nf[nb + i] = (int16_t)(nf[nb + i] + nf[nb + i - 1]);
    } else {
      for (int i = 0; i < nb; i++) nf[nb + i] = (int16_t)(nf[nb + i] + nf[i]);
    }
  }
  for (int i = 0; i < nb * ne; i++) { /* :345-376 */
    if (nf[i] > 35) nf[i] = 35;
    else if (nf[i] < 0) nf[i] = 0;
  }
  const int offset = nb * (ne - 1);
  if (offset < 0 || offset >= XAAC_SBR_MAX_NOISE_VALUES) return -1;
  memcpy(p->prev_noise_level, nf + offset, sizeof(int16_t) * (size_t)nb);
  if (enh || f->coupling_mode != XS_COUPLING_BAL)
    for (int i = 0; i < nb * ne; i++) {
      f->flt_noise_floor[i] = (float)nf[i]; /* the float tools' copy: the limited level itself (:471-476) */
      nf[i] = (int16_t)(0x4000 + ((6 + 1 + kNoiseExpOffset - nf[i]) & kMaskExp));
    }
  return 0;
}

void dequant_coupled(const XsHeader *h, XsFrameData *l, XsFrameData *r) { /* :516-584 */
  for (int i = 0; i < l->num_env_sfac; i++) {
    const int16_t rm = (int16_t)(r->env_sf[i] & kMaskM);
    const int16_t re = (int16_t)((r->env_sf[i] & kMaskExp) - (18 + kNrgExpOffset));
    const int16_t lm = (int16_t)(l->env_sf[i] & kMaskM);
    const int16_t le = (int16_t)((l->env_sf[i] & kMaskExp) - kNrgExpOffset);
    int16_t r1m, r1e, nrm;
    mant_exp_add(rm, re, 0x4000, 1, &r1m, &r1e);
    int16_t nre = (int16_t)mant_div(lm, r1m, &nrm);
    nre = (int16_t)(nre + le - r1e + 2);
    const int16_t nlm = (int16_t)(((int32_t)rm * (int32_t)nrm) >> 15);
    const int16_t nle = add16(re, nre);
    r->env_sf[i] = (int16_t)(((nrm + kRounding) & kMaskM) + ((nre + kNrgExpOffset) & kMaskExp));
    l->env_sf[i] = (int16_t)(((nlm + kRounding) & kMaskM) + ((nle + kNrgExpOffset) & kMaskExp));
  }
  const int n = h->num_nf_bands * l->fi.num_noise_env;
  for (int i = 0; i < n; i++) {
    const int16_t le = (int16_t)((l->noise_floor[i] & kMaskExp) - kNoiseExpOffset);
    const int16_t re = (int16_t)(r->noise_floor[i] - 12);
    int16_t r1m, r1e, nrm;
    mant_exp_add(0x4000, (int16_t)(1 + re), 0x4000, 1, &r1m, &r1e);
    int16_t nre = (int16_t)mant_div(0x4000, r1m, &nrm);
    nre = (int16_t)(nre + le - r1e + 2);
    const int16_t nle = add16(nre, re);
    r->noise_floor[i] = (int16_t)(((nrm + kRounding) & kMaskM) + ((nre + kNoiseExpOffset) & kMaskExp));
    l->noise_floor[i] = (int16_t)(((nrm + kRounding) & kMaskM) + ((nle + kNoiseExpOffset) & kMaskExp));
  }
}

/* the float tools' scale factors: :52-72 (one channel) and :586-626 (a coupled pair) */
void dequant_float(const XsHeader *h, XsFrameData *f) {
  const float a = f->amp_res ? 1.0f : 0.5f;
  for (int i = 0; i < f->num_env_sfac; i++) f->flt_env_sf[i] = (float)(pow(2, f->flt_env_sf[i] * a) * 64);
  const int n = h->num_nf_bands * f->fi.num_noise_env;
  for (int i = 0; i < n; i++) {
    float t = f->flt_noise_floor[i];
    t = 6.0f - t;
    f->flt_noise_floor[i] = (float)pow(2.0f, t);
  }
}
void dequant_float_coupled(const XsHeader *h, XsFrameData *l, XsFrameData *r) {
  static const float pan_offset[2] = {24.0f, 12.0f}, a_arr[2] = {0.5f, 1.0f};
  const int amp = l->amp_res;
  const float a = a_arr[amp];
  for (int i = 0; i < l->num_env_sfac; i++) {
    const float tl = l->flt_env_sf[i], tr = r->flt_env_sf[i];
    l->flt_env_sf[i] = (float)(64 * (pow(2, tl * a + 1) / (1 + pow(2, (pan_offset[amp] - tr) * a))));
    r->flt_env_sf[i] = (float)(64 * (pow(2, tl * a + 1) / (1 + pow(2, (tr - pan_offset[amp]) * a))));
  }
  const int n = h->num_nf_bands * r->fi.num_noise_env;
  for (int i = 0; i < n; i++) {
    const float tl = l->flt_noise_floor[i], tr = r->flt_noise_floor[i];
    l->flt_noise_floor[i] = (float)(pow(2, 6.0f - tl + 1) / (1 + pow(2, pan_offset[1] - tr)));
    r->flt_noise_floor[i] = (float)(pow(2, 6.0f - tl + 1) / (1 + pow(2, tr - pan_offset[1])));
  }
}

int decode_sbr_data(XsDecoder *d, int two) { /* ixheaacd_dec_sbrdata :628-725 */
  XsHeader *h = &d->hdr;
  const int enh = d->enh;
  if (decode_envelope(h, &d->fd[0], &d->prev[0], &d->prev[1], enh)) return -1;
  if (decode_noise(h, &d->fd[0], &d->prev[0], enh)) return -1;
  if (enh && !d->fd[0].coupling_mode) dequant_float(h, &d->fd[0]);
  if (two) {
    const int err = h->err_flag;
    if (decode_envelope(h, &d->fd[1], &d->prev[1], &d->prev[0], enh)) return -1;
    if (decode_noise(h, &d->fd[1], &d->prev[1], enh)) return -1;
    if (enh && !d->fd[1].coupling_mode) dequant_float(h, &d->fd[1]);
    if (!enh && !err && h->err_flag)
      if (decode_envelope(h, &d->fd[0], &d->prev[0], &d->prev[1], enh)) return -1;
    if (d->fd[0].coupling_mode) {
      dequant_coupled(h, &d->fd[0], &d->fd[1]);
      if (enh) dequant_float_coupled(h, &d->fd[0], &d->fd[1]);
    }
  }
  return 0;
}

/* ---- PS index decoding: ps_bitdec.c:79-282 ------------------------------------------------------------------------------- */
inline int16_t div2(int v) { return (int16_t)(v < 0 ? -((-v) >> 1) : v >> 1); }
inline int16_t div3(int v) {
  const int neg = v < 0;
  if (neg) v = -v;
  int16_t t = (int16_t)(v << 2);
  t = (int16_t)(((int32_t)t * 0x2aab) >> 15);
  const int16_t r = (int16_t)(t >> 2);
  return neg ? (int16_t)-r : r;
}
void map_34_to_20(int16_t *p) { /* sbrdec_lpfuncs.c:561-583 */
  p[0] = div3(p[0] + p[0] + p[1]);
  p[1] = div3(p[1] + p[2] + p[2]);
  p[2] = div3(p[3] + p[3] + p[4]);
  p[3] = div3(p[4] + p[5] + p[5]);
  p[4] = div2(p[6] + p[7]);
  p[5] = div2(p[8] + p[9]);
  p[6] = p[10];
  p[7] = p[11];
  p[8] = div2(p[12] + p[13]);
  p[9] = div2(p[14] + p[15]);
  p[10] = p[16];
  p[11] = p[17];
  p[12] = p[18];
  p[13] = p[19];
  p[14] = div2(p[20] + p[21]);
  p[15] = div2(p[22] + p[23]);
  p[16] = div2(p[24] + p[25]);
  p[17] = div2(p[26] + p[27]);
  p[18] = div2(div2(p[28] + p[29] + p[30] + p[31]));
  p[19] = div2(p[32] + p[33]);
}
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void decode_ps(XsPs *ps) {
  static const int num_bands[3] = {10, 20, 34};
  const int iid_step = ps->iid_mode ? 1 : 2, icc_step = ps->icc_mode ? 1 : 2;
  const int levels = ps->iid_quant ? 15 : 7;
  if (!ps->data_present) ps->num_env = 0;
  for (int e = 0; e < ps->num_env; e++) {
    const int16_t *iid_prev = e == 0 ? ps->iid_prev : ps->iid_par[e - 1];
    const int16_t *icc_prev = e == 0 ? ps->icc_prev : ps->icc_par[e - 1];
    int16_t *iid = ps->iid_par[e], *icc = ps->icc_par[e];
    const int ni = num_bands[ps->iid_mode], nc = num_bands[ps->icc_mode];
    if (ps->enable_iid) {
      if (ps->iid_dt[e]) {
        for (int i = 0; i < ni; i++) iid[i] = (int16_t)clampi((int16_t)(iid_prev[i * iid_step] + iid[i]), -levels, levels);
      } else {
        iid[0] = (int16_t)clampi(iid[0], -levels, levels);
        for (int i = 1; i < ni; i++) iid[i] = (int16_t)clampi((int16_t)(iid[i - 1] + iid[i]), -levels, levels);
      }
    } else {
      memset(iid, 0, sizeof(int16_t) * (size_t)ni);
    }
    if (iid_step == 2)
      for (int i = ni * 2 - 1; i != 0; i--) iid[i] = iid[i >> 1];
    if (ps->enable_icc) {
      if (ps->icc_dt[e]) {
        for (int i = 0; i < nc; i++) icc[i] = (int16_t)clampi((int16_t)(icc_prev[i * icc_step] + icc[i]), 0, 7);
      } else {
        icc[0] = (int16_t)clampi(icc[0], 0, 7);
        for (int i = 1; i < nc; i++) icc[i] = (int16_t)clampi((int16_t)(icc[i - 1] + icc[i]), 0, 7);
      }
    } else {
      memset(icc, 0, sizeof(int16_t) * (size_t)nc);
    }
    if (icc_step == 2)
      for (int i = nc * 2 - 1; i != 0; i--) icc[i] = icc[i >> 1];
  }
  if (ps->num_env == 0) {
    ps->num_env = 1;
    if (ps->enable_iid) memcpy(ps->iid_par[0], ps->iid_prev, sizeof(ps->iid_prev));
    else memset(ps->iid_par[0], 0, sizeof(ps->iid_prev));
    if (ps->enable_icc) memcpy(ps->icc_par[0], ps->icc_prev, sizeof(ps->icc_prev));
    else memset(ps->icc_par[0], 0, sizeof(ps->icc_prev));
  }
  memcpy(ps->iid_prev, ps->iid_par[ps->num_env - 1], sizeof(ps->iid_prev));
  memcpy(ps->icc_prev, ps->icc_par[ps->num_env - 1], sizeof(ps->icc_prev));
  ps->data_present = 0;
  const int cols = 32;
  if (ps->frame_class == 0) {
    const int shift = ps->num_env == 2 ? 1 : (ps->num_env == 4 ? 2 : 0);
    ps->border_position[0] = 0;
    int count = 0;
    for (int e = 1; e < ps->num_env; e++) {
      count += cols;
      ps->border_position[e] = (int16_t)(count >> shift);
    }
    ps->border_position[ps->num_env] = cols;
  } else {
    ps->border_position[0] = 0;
    if (ps->border_position[ps->num_env] < cols) {
      ps->num_env++;
      ps->border_position[ps->num_env] = cols;
      memcpy(ps->iid_par[ps->num_env - 1], ps->iid_par[ps->num_env - 2], sizeof(ps->iid_prev));
      memcpy(ps->icc_par[ps->num_env - 1], ps->icc_par[ps->num_env - 2], sizeof(ps->icc_prev));
    }
    for (int e = 1; e < ps->num_env; e++) {
      int thr = cols - (ps->num_env - e);
      if (ps->border_position[e] > thr) {
        ps->border_position[e] = (int16_t)thr;
      } else {
        thr = ps->border_position[e - 1] + 1;
        if (ps->border_position[e] < thr) ps->border_position[e] = (int16_t)thr;
      }
    }
  }
  for (int e = 0; e < ps->num_env; e++) {
    if (ps->iid_mode == 2) map_34_to_20(ps->iid_par[e]);
    if (ps->icc_mode == 2) map_34_to_20(ps->icc_par[e]);
  }
}

/* ---- CRC of the payload: sbr_crc.c:26-97 --------------------------------------------------------------------------------- */
int crc_ok(XhBits *br, int crc_bits) {
  const uint32_t want = br->get(10);
  const int avail = (int)br->left();
  if (avail <= 0) return 0;
  const int n = crc_bits > avail ? avail : crc_bits;
  XhBits local = *br;
  uint16_t state = 0;
  for (int i = 0; i < n; i++) {
    const int bit = local.get1() ^ ((state >> 9) & 1);
    state = (uint16_t)(state << 1);
    if (bit) state ^= 0x0233;
  }
  return (uint32_t)(state & 0x3ff) == want;
}

void prepare_upsampling(XsDecoder *d, XsFrameResult *res) { /* sbrdecoder.c:254-276 */
  d->hdr.sub_band_start = 32;
  d->hdr.sub_band_end = 64;
  d->hdr.sync_state = XS_UPSAMPLING;
  res->upsampling = 1;
}

void export_header(const XsHeader *h, xaac_sbr_header *o) { /* oracle/ref_convert.h: to_header */
  memset(o, 0, sizeof(*o));
  o->num_time_slots = kTimeSlots;
  o->time_step = 2;
  o->channel_mode = (int16_t)h->channel_mode;
  o->limiter_gains = (int16_t)h->limiter_gains;
  o->interpol_freq = (int16_t)h->interpol_freq;
  o->smoothing_mode = (int16_t)h->smoothing_mode;
  o->num_sf_bands[0] = h->num_sf_bands[0], o->num_sf_bands[1] = h->num_sf_bands[1];
  o->num_nf_bands = h->num_nf_bands;
  o->sub_band_start = h->sub_band_start;
  o->sub_band_end = h->sub_band_end;
  o->num_lf_bands = h->num_lf_bands;
  o->num_if_bands = h->num_if_bands;
  memcpy(o->freq_band_tbl_lim, h->tbl_lim, sizeof(o->freq_band_tbl_lim));
  memcpy(o->freq_band_tbl_lo, h->tbl_lo, sizeof(o->freq_band_tbl_lo));
  memcpy(o->freq_band_tbl_hi, h->tbl_hi, sizeof(o->freq_band_tbl_hi));
  memcpy(o->freq_band_tbl_noise, h->tbl_noise, sizeof(o->freq_band_tbl_noise));
  o->num_columns = 32;
  o->num_patches = h->num_patches;
  o->start_patch = h->start_patch;
  o->stop_patch = h->stop_patch;
  memcpy(o->bw_borders, h->bw_borders, sizeof(o->bw_borders));
  memcpy(o->patch, h->patch, sizeof(o->patch));
}

void export_frame(const XsFrameData *f, int apply, xaac_sbr_frame *o) { /* to_frame */
  memset(o, 0, sizeof(*o));
  o->num_env = f->fi.num_env, o->transient_env = f->fi.transient_env, o->num_noise_env = f->fi.num_noise_env;
  o->frame_class = f->fi.frame_class;
  memcpy(o->border_vec, f->fi.border_vec, sizeof(o->border_vec));
  memcpy(o->freq_res, f->fi.freq_res, sizeof(o->freq_res));
  memcpy(o->noise_border_vec, f->fi.noise_border_vec, sizeof(o->noise_border_vec));
  o->amp_res = f->amp_res;
  o->apply_processing = (int16_t)apply;
  o->coupling_mode = f->coupling_mode;
  o->max_qmf_subband_aac = f->max_qmf_subband_aac;
  memcpy(o->sbr_invf_mode, f->invf_mode, sizeof(o->sbr_invf_mode));
  memcpy(o->add_harmonics, f->add_harmonics, sizeof(o->add_harmonics));
  memcpy(o->int_env_sf_arr, f->env_sf, sizeof(o->int_env_sf_arr));
  memcpy(o->int_noise_floor, f->noise_floor, sizeof(o->int_noise_floor));
}

}  // namespace

void xs_init(XsDecoder *d, int core_sampling_rate, int core_channels, int ps_enable, int enh) {
  static const bool luts_built = (build_luts(), true);
  (void)luts_built;
  memset(d, 0, sizeof(*d));
  d->core_channels = core_channels;
  d->ps_enable = ps_enable;
  d->enh = enh;
  d->qmf_sb_prev = 64;
  { /* str_sbr_default_header (ixheaacd_sbr_rom.c:2008-2030), copied into a new stream's header (sbrdec_initfuncs.c:556) */
    XsHeader *h = &d->hdr;
    h->amp_res = 1, h->start_freq = 15, h->stop_freq = 6, h->xover_band = 0, h->freq_scale = 2, h->alter_scale = 1, h->noise_bands = 2;
    h->limiter_bands = 2, h->limiter_gains = 2, h->interpol_freq = 1, h->smoothing_mode = 1;
  }
  d->hdr.out_sampling_freq = 2 * core_sampling_rate;
  d->hdr.sync_state = XS_NOT_INITIALIZED;
  for (int c = 0; c < 2; c++) d->prev[c].end_position = kTimeSlots; /* sbrdec_initfuncs.c:884-898 */
  for (int c = 0; c < 2; c++) d->fd[c].fi.transient_env = 0;
}

int xs_decode_frame(XsDecoder *d, const uint8_t *payload, int bytes, int ext_type, xaac_sbr_header *header,
                    xaac_sbr_frame frame[2], xaac_ps_frame *ps_frame, XsFrameResult *res) {
  XsHeader *h = &d->hdr;
  memset(res, 0, sizeof(*res));
  const int num_channels = d->core_channels;
  int frame_status = 1, stereo = 0, err = 0;
  const int prev_ps = d->ps_enable && h->channel_mode == XS_PS_STEREO;
  const int prev_stereo = h->channel_mode == XS_SBR_STEREO;
  const int initial_sync = h->sync_state;
  h->err_flag_prev = h->err_flag;
  const int lr1 = d->ps_enable ? 2 : num_channels;
  uint8_t late[sizeof(d->prev_payload)];
  int skip_element = 0;
  if (bytes == 0) {
    frame_status = 0;
    h->sync_state = XS_UPSAMPLING;
  } else if (d->enh) { /* the payload decoded now is the previous frame's; this one waits (sbrdecoder.c:479-493) */
    const int nb = d->prev_bytes, nt = d->prev_ext_type;
    memcpy(late, d->prev_payload, sizeof(late));
    memcpy(d->prev_payload, payload, (size_t)bytes);
    d->prev_bytes = bytes, d->prev_ext_type = ext_type;
    payload = late, bytes = nb, ext_type = nt;
    if (bytes == 0) skip_element = 1; /* `continue`: nothing of the element is looked at */
  }
  if (bytes != 0) {
    XhBits br(payload, (size_t)bytes);
    stereo = num_channels == 2; /* the payload of a CPE (the element types follow the core channels in this scope) */
    br.get(4);                  /* the nibble behind the extension type (sbrdecoder.c:495) */
    if (ext_type == 14) {
      const int crc_bits = ((bytes - 1) << 3) + (4 - 10);
      frame_status = crc_bits < 0 ? 0 : crc_ok(&br, crc_bits);
    }
    int header_flag = br.get1();
    if (header_flag) {
      header_flag = read_header(h, &br, stereo);
      if (header_flag == SBR_RESET) {
        err = calc_freq_tables(h);
        if (!err) {
          for (int lr = 0; lr < lr1; lr++) {
            d->fd[lr].reset_flag = 1;
            if (h->sync_state == XS_NOT_INITIALIZED)
              d->fd[lr].patching_mode = 1, d->fd[lr].over_sampling = 0, d->fd[lr].pitch_in_bins = 0, h->pre_flatten = 0;
          }
          int e2 = reset_hf_generator(h);
          if (e2 < 0) return -1;
          err |= e2;
          derive_lim_bands(h);
          res->reset = 1;
          res->reset_channels = lr1;
          d->reset_pitch = d->fd[0].pitch_in_bins; /* what ixheaacd_sbr_dec_reset is handed: the element's first channel's
                                                      pitch of the payload before this one (sbrdecoder.c:547-550) */
        }
        if (err == 0) h->sync_state = XS_ACTIVE;
      }
    }
    if (err || h->sync_state == XS_NOT_INITIALIZED) {
      prepare_upsampling(d, res);
      if (err) return -1;
    }
    if (frame_status && h->sync_state == XS_ACTIVE) {
      if (stereo) frame_status = read_cpe(h, d->fd, &br, d->enh);
      else frame_status = read_sce(h, &d->fd[0], d->ps_enable ? &d->ps : nullptr, &br, d->enh);
      if (frame_status < 0) return -1;
      const int read = (int)br.pos;
      if (read > (bytes << 3) || read < (bytes << 3) - 8) frame_status = 0;
    }
  }
  if (!frame_status || h->sync_state != XS_ACTIVE || h->err_flag) {
    h->err_flag = 1;
    stereo = num_channels == 2;
    if (h->channel_mode == 0) h->channel_mode = stereo ? XS_SBR_STEREO : XS_SBR_MONO;
  }
  if (!stereo) d->fd[0].coupling_mode = d->fd[1].coupling_mode = XS_COUPLING_OFF;
  if (h->sync_state == XS_NOT_INITIALIZED) prepare_upsampling(d, res);
  int ps_flag = 0;
  if (h->sync_state == XS_ACTIVE) {
    if (decode_sbr_data(d, stereo)) return -1;
    if (h->channel_mode == XS_PS_STEREO) {
      decode_ps(&d->ps);
      ps_flag = 1;
    }
    d->fd[0].max_qmf_subband_aac = h->sub_band_start;
    if (stereo) d->fd[1].max_qmf_subband_aac = h->sub_band_start;
  }
  if (initial_sync == XS_NOT_INITIALIZED && h->err_flag) h->sync_state = XS_NOT_INITIALIZED;
  (void)skip_element;
  res->apply = h->sync_state == XS_ACTIVE;
  res->stereo = stereo;
  res->ps = h->channel_mode == XS_PS_STEREO;
  res->ps_start = !prev_stereo && !prev_ps && ps_flag;
  res->frame_ok = frame_status;
  export_header(h, header);
  export_frame(&d->fd[0], res->apply, &frame[0]);
  export_frame(&d->fd[1], res->apply, &frame[1]);
  if (ps_frame) {
    memset(ps_frame, 0, sizeof(*ps_frame));
    ps_frame->iid_quant = (int16_t)d->ps.iid_quant;
    ps_frame->freq_res_ipd = (int16_t)d->ps.freq_res_ipd;
    ps_frame->num_env = (int16_t)d->ps.num_env;
    memcpy(ps_frame->border_position, d->ps.border_position, sizeof(ps_frame->border_position));
    memcpy(ps_frame->iid_par_table, d->ps.iid_par, sizeof(ps_frame->iid_par_table));
    memcpy(ps_frame->icc_par_table, d->ps.icc_par, sizeof(ps_frame->icc_par_table));
  }
  return 0;
}

int xs_hbe_k_start(int start_band) { return start_band >= 0 && start_band <= 32 ? xh_start_subband2kl[start_band] : -1; }

void xs_export_esbr_side(const XsDecoder *d, int c, xaac_esbr_side *o) { /* oracle/ref_convert.h: to_esbr_side */
  const XsHeader *h = &d->hdr;
  const XsFrameData *f = &d->fd[c];
  memset(o, 0, sizeof(*o));
  o->out_sampling_freq = h->out_sampling_freq;
  o->limiter_bands = (int16_t)h->limiter_bands;
  o->num_mf_bands = h->num_mf_bands;
  memcpy(o->f_master_tbl, h->f_master, sizeof(o->f_master_tbl));
  o->qmf_sb_prev = (int16_t)d->qmf_sb_prev_frame;
  o->reset_flag = (int16_t)f->reset_flag_frame;
  o->harmonic_sbr = (int16_t)((f->patching_mode == 0 ? XAAC_ESBR_HARMONIC : 0) | (h->pre_flatten ? XAAC_ESBR_PRE_FLATTEN : 0) |
                              (f->over_sampling ? XAAC_ESBR_OVERSAMPLING : 0));
  memcpy(o->sbr_invf_mode_prev, f->invf_mode_prev, sizeof(o->sbr_invf_mode_prev));
  memcpy(o->flt_env_sf_arr, f->flt_env_sf, sizeof(o->flt_env_sf_arr));
  memcpy(o->flt_noise_floor, f->flt_noise_floor, sizeof(o->flt_noise_floor));
  o->pitch_in_bins = f->pitch_in_bins;
}

void xs_frame_done(XsDecoder *d, const XsFrameResult *res) {
  /* qmf_sb_prev follows the start band behind the frame's ixheaacd_sbr_dec calls (sbrdecoder.c:915-990, the calls with a DRC
     handle): channel 0 of a mono stream, channel 1 of a pair (both write the one table the channels share) */
  d->qmf_sb_prev_frame = d->qmf_sb_prev;
  for (int c = 0; c < 2; c++) d->fd[c].reset_flag_frame = d->fd[c].reset_flag;
  if (d->enh) /* the float synthesis stage takes the flag down behind every frame (sbr_dec.c:659) */
    for (int c = 0; c < (d->core_channels == 2 ? 2 : 1); c++) d->fd[c].reset_flag = 0;
  if (d->enh && res->apply && (d->core_channels != 2 || res->stereo)) d->qmf_sb_prev = d->hdr.sub_band_start;
  if (!res->apply) return;
  const int n = (res->stereo && d->core_channels == 2) ? 2 : 1;
  for (int c = 0; c < n; c++) {
    const XsFrameData &f = d->fd[c];
    XsPrevData &p = d->prev[c];
    memcpy(p.invf_mode, f.invf_mode, sizeof(int32_t) * (size_t)d->hdr.num_if_bands);
    p.coupling_mode = f.coupling_mode;
    p.max_qmf_subband_aac = f.max_qmf_subband_aac;
    p.end_position = f.fi.border_vec[f.fi.num_env];
    p.amp_res = f.amp_res;
  }
}
