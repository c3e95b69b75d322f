/*
 * sbr_core_reg.h -- the HQ (complex) SBR core with the QMF matrix in REGISTERS: lane = QMF band, every lane owns its
 * band's column of 2 LPC-history rows + 38 slots, real and imaginary part side by side (80 lane-vector elements = 80
 * VGPRs on the GPU).  Same arithmetic as sbr_core.h's matrix-in-memory form -- the envelope adjuster's gain mathematics
 * (xs_subband_gain_meta .. xs_noiselimiting, xs_erg_to_amplitude_hq) IS that code; what is restated here is everything
 * that touches the matrix:
 *   xs_headroom / xs_adjust / xs_clear / xs_lpc_save     env_calc.c:1159 / :1099, sbr_dec.c:1221
 *   xs_hf_generator_hq                                    lpp_tran.c:956 (covariances :372, coefficients :1041, patching :102 / :1203)
 *   xs_energy_per_subband                                 env_calc.c:1211
 *   xs_adapt_noise_gain_hq                                env_calc.c:479 with env_dec.c:845 and env_calc.c:1759 / :1827
 * Why: in memory (LDS) the 40 x 128-word matrix is 15-20 KB per stream, which caps a CU at eight resident waves, and
 * every pass over the slots is a chain of dependent LDS round trips (read, compute, write) -- the kernel was bound by
 * that latency at two waves per SIMD.  In registers a pass over the slots is straight-line VALU code
 * with no memory operation at all, and LDS only holds the side info and the state tail (6 KB per wave).
 * The price: a register array can only be indexed by constants, so every walk over slots is a fully unrolled loop over
 * ALL slots whose body is guarded by the (wave-uniform) envelope / frame borders, and the one place where a band reads
 * another band's column -- patching: high band <- low band -- is a lane gather (ds_bpermute) per row.
 *
 * Host: the same source with XsLv = 64-element arrays runs all "lanes" sequentially (cx.n = 1); oracle/oracle_sbr.cpp
 * can route its HQ core through it and tests/test_core_reg_cpu.py holds the two forms against each other on the
 * reference's captured frames and fuzzed chains -- before the GPU sees the code.
 */
#ifndef XAAC_SBR_CORE_REG_H
#define XAAC_SBR_CORE_REG_H

#include "sbr_core.h"

#define XS_REG_ROWS 40 /* row r = slot r - 2 */

struct XsQmfReg {
  static constexpr int HQ = 1;
  static constexpr int NB = 64;
  mutable XsLv re[XS_REG_ROWS], im[XS_REG_ROWS];
};

/* every slot, as an unrolled loop with constant indices */
#define XS_SLOTS(l) XS_UNROLL for (int l = 0; l < 38; l++)
/* On the GPU the unrolled walks are kept in slot order: XS_BR() keeps a guarded slot body a real (wave-uniform) branch
   instead of 80 selects on 64-bit masks, XS_PIN(v) makes a value the walk consumes opaque at that point so that the products
   of all 38 slots are not computed up front (the compiler would otherwise trade 150 registers for that) */
#if defined(__HIP_DEVICE_COMPILE__)
#define XS_BR() asm volatile("")
#define XS_PIN(v) asm volatile("" : "+v"(v))
#else
#define XS_BR()
#define XS_PIN(v)
#endif

/* LA, LB: slots the call can touch at all (constants: the loop is unrolled over them only) */
template <int LA = 0, int LB = 38>
FX_HD int xs_headroom(const XsCx &cx, const XsQmfReg &x, int b0, int b1, int s0, int s1) {
  int32_t m = 1;
  XS_LANES(k, 0, 64) {
    if (k >= b0 && k < b1) {
      XS_UNROLL for (int l = LA; l < LB; l++) {
        if (l >= s0 && l < s1) {
          XS_BR();
          m |= fx_abs_nrm(x.re[l + 2].own(k)) | fx_abs_nrm(x.im[l + 2].own(k));
        }
      }
    }
  }
  return xs_pnorm32(cx.wave_or(m));
}

template <int LA = 0, int LB = 38>
FX_HD void xs_adjust(const XsCx &cx, const XsQmfReg &x, int b0, int b1, int s0, int s1, int shift) {
  if (shift == 0) return;
  if (shift > 31) shift = 31;
  if (shift < -31) shift = -31;
  XS_LANES(k, 0, 64) {
    if (k >= b0 && k < b1) {
      XS_UNROLL for (int l = LA; l < LB; l++) {
        if (l >= s0 && l < s1) {
          XS_BR();
          x.re[l + 2].own(k) = shift > 0 ? fx_shlw(x.re[l + 2].own(k), shift) : (x.re[l + 2].own(k) >> -shift);
          x.im[l + 2].own(k) = shift > 0 ? fx_shlw(x.im[l + 2].own(k), shift) : (x.im[l + 2].own(k) >> -shift);
        }
      }
    }
  }
}

template <int LA = 0, int LB = 38>
FX_HD void xs_clear(const XsCx &cx, const XsQmfReg &x, int b0, int b1, int s0, int s1) {
  XS_LANES(k, 0, 64) {
    if (k >= b0 && k < b1) {
      XS_UNROLL for (int l = LA; l < LB; l++) {
        if (l >= s0 && l < s1) {
          XS_BR();
          x.re[l + 2].own(k) = 0;
          x.im[l + 2].own(k) = 0;
        }
      }
    }
  }
}

/* sbrdec_lpfuncs.c:453 on the register matrix: xs_rescale_x_overlap of sbr_core.h with its slot walks unrolled over the six
   overlap slots only (start_slot <= 6: xs_side_info_bad bounds prev_end_position) */
template <class ST>
FX_HD void xs_rescale_x_overlap(const XsCx &cx, const xaac_sbr_header *h, const xaac_sbr_frame *f, ST *st, const XsQmfReg &x) {
  const int old_lsb = cx.uni(st->prev_max_qmf_subband_aac);
  const int start_slot = cx.uni(h->time_step) * (cx.uni(st->prev_end_position) - cx.uni(h->num_time_slots));
  const int new_lsb = cx.uni(f->max_qmf_subband_aac);
  const int ov_hb = cx.uni(st->ov_hb_scale), ov_lb = cx.uni(st->ov_lb_scale), syn_usb = cx.uni(st->syn_usb);
  cx.sync();
  XS_ONE {
    st->codec_usb = (int16_t)new_lsb;
    st->syn_lsb = (int16_t)new_lsb;
  }
  int b0 = old_lsb < new_lsb ? old_lsb : new_lsb, b1 = old_lsb < new_lsb ? new_lsb : old_lsb;
  if (new_lsb == old_lsb || old_lsb <= 0) {
    cx.sync();
    return;
  }
  xs_clear<0, 6>(cx, x, old_lsb, new_lsb, start_slot, 6);
  int source, target, t_lsb, t_usb;
  if (new_lsb > old_lsb) {
    source = ov_hb;
    target = ov_lb;
    t_lsb = 0;
    t_usb = old_lsb;
  } else {
    source = ov_lb;
    target = ov_hb;
    t_lsb = old_lsb;
    t_usb = syn_usb;
  }
  cx.sync();
  const int reserve = xs_headroom<0, 6>(cx, x, b0, b1, 0, start_slot);
  xs_adjust<0, 6>(cx, x, b0, b1, 0, start_slot, reserve);
  source += reserve;
  int delta = target - source;
  if (delta > 0) {
    delta = -delta;
    b0 = t_lsb;
    b1 = t_usb;
    XS_ONE {
      if (new_lsb > old_lsb)
        st->ov_lb_scale = (int16_t)source;
      else
        st->ov_hb_scale = (int16_t)source;
    }
  }
  cx.sync();
  xs_adjust<0, 6>(cx, x, b0, b1, 0, start_slot, delta);
  cx.sync();
}

template <class ST>
FX_HD void xs_lpc_save(const XsCx &cx, ST *st, const XsQmfReg &x, int usb) {
  XS_LANES(k, 0, usb) { /* the same expressions as the matrix-in-memory form, whatever usb is */
    st->lpc_real[0][k] = x.re[32].own(k);
    st->lpc_real[1][k] = x.re[33].own(k);
    st->lpc_imag[0][k] = x.im[32].own(k);
    st->lpc_imag[1][k] = x.im[33].own(k);
  }
}

/* lpp_tran.c:372: the eight covariance sums of low band k, all 38 slots in the lane (see xs_covariance_hq) */
FX_HD void xs_covariance_hq_reg(const XsQmfReg &x, int k, XsCovHq *c) {
  int32_t p01 = 0, p01i = 0, p02 = 0, p02i = 0, p11 = 0, p12 = 0, p12i = 0, p22 = 0;
  int32_t r2 = fx_shr(x.re[0].own(k), 3), i2 = fx_shr(x.im[0].own(k), 3); /* x[n-2] */
  int32_t r1 = fx_shr(x.re[1].own(k), 3), i1 = fx_shr(x.im[1].own(k), 3); /* x[n-1] */
  p22 = fx_add(xs_mul_hi16(r2, r2), xs_mul_hi16(i2, i2));
  p12 = fx_add(xs_mul_hi16(r1, r2), xs_mul_hi16(i1, i2));
  p12i = fx_sub(xs_mul_hi16(i1, r2), xs_mul_hi16(r1, i2));
  XS_SLOTS(n) {
    int32_t r0 = fx_shr(x.re[n + 2].own(k), 3), i0 = fx_shr(x.im[n + 2].own(k), 3);
    XS_PIN(r0);
    XS_PIN(i0);
    const int32_t t01 = fx_add(xs_mul_hi16(r0, r1), xs_mul_hi16(i0, i1));
    const int32_t t01i = fx_sub(xs_mul_hi16(i0, r1), xs_mul_hi16(r0, i1));
    const int32_t e1 = fx_add(xs_mul_hi16(r1, r1), xs_mul_hi16(i1, i1)); /* |x[n-1]|^2 */
    p01 = fx_add(p01, t01);
    p01i = fx_add(p01i, t01i);
    p02 = fx_add(p02, fx_add(xs_mul_hi16(r0, r2), xs_mul_hi16(i0, i2)));
    p02i = fx_add(p02i, fx_sub(xs_mul_hi16(i0, r2), xs_mul_hi16(r0, i2)));
    p11 = fx_add(p11, e1);
    if (n < 37) { /* the shifted sums stop one sample earlier */
      p12 = fx_add(p12, t01);
      p12i = fx_add(p12i, t01i);
      p22 = fx_add(p22, e1);
    }
    r2 = r1;
    i2 = i1;
    r1 = r0;
    i1 = i0;
  }
  c->phi_11 = p11;
  c->phi_22 = p22;
  c->phi_01 = p01;
  c->phi_02 = p02;
  c->phi_12 = p12;
  c->phi_01_im = p01i;
  c->phi_02_im = p02i;
  c->phi_12_im = p12i;
}

/* lpp_tran.c:956 on the register matrix.  Writes bw_array_prev. */
template <class ST>
FX_HD void xs_hf_generator_hq(const XsCx &cx, const xaac_sbr_header *h, ST *st, const XsQmfReg &x, XsWork *w,
                              int start_idx, int last_slot_offset, int max_qmf_subband, const int32_t *invf_mode,
                              const int32_t *invf_mode_prev) {
  const int num_patches = cx.uni(h->num_patches);
  const int stop_idx = cx.uni(h->num_columns) + last_slot_offset;
  XS_PAR(i, 0, XAAC_SBR_MAX_PATCHES) w->bw_array[i] = 0;
  cx.sync();
  xs_invfilt_level_emphasis(cx, st->bw_array_prev, h->num_if_bands, invf_mode, invf_mode_prev, w->bw_array);
  const int actual_stop = cx.uni(
      (int16_t)(h->patch[num_patches - 1].dst_start_band + h->patch[num_patches - 1].num_bands_in_patch));
  xs_clear(cx, x, actual_stop, 64, start_idx, stop_idx);
  const int start_patch = cx.uni(h->start_patch), stop_patch = cx.uni(h->stop_patch);
  XS_LANES(k, 0, 64) {
    if (k >= start_patch && k < stop_patch) {
      x.re[0].own(k) = st->lpc_real[0][k];
      x.re[1].own(k) = st->lpc_real[1][k];
      x.im[0].own(k) = st->lpc_imag[0][k];
      x.im[1].own(k) = st->lpc_imag[1][k];
    }
  }
  cx.sync();
  XS_T(12);
  XsLv al01, al23, src;
  al01.fill(0);
  al23.fill(0);
  XS_LANES(lb, 0, 64) {
    if (lb >= start_patch && lb < stop_patch) {
      XsCovHq c;
      xs_covariance_hq_reg(x, lb, &c);
      int16_t alpha[4];
      xs_lpc_coeffs_hq(&c, alpha);
      al01.own(lb) = xs_me(alpha[0], alpha[1]);
      al23.own(lb) = xs_me(alpha[2], alpha[3]);
    }
  }
  XS_T(13);
  /* the low band behind each high band (xs_hf_generator_hq in sbr_core.h: largest low band, later patch among equals) */
  src.fill(-1);
  XS_LANES(hb, 0, 64) {
    int best = -1;
    for (int patch = 0; patch < num_patches; patch++) {
      const xaac_sbr_patch *pp = &h->patch[patch];
      const int lb = hb - pp->dst_end_band;
      if (lb < pp->src_start_band || lb >= pp->src_end_band || lb < start_patch || lb >= stop_patch) continue;
      if (hb < max_qmf_subband) continue;
      if (lb >= best) best = lb;
    }
    src.own(hb) = best;
  }
  const XsLv a01 = al01.gather(src), a23 = al23.gather(src);
  /* the lane's filter (lpp_tran.c:1203-1250, xs_patch_band_hq): chirped coefficients, or none where bw is not positive --
     with all four at zero the filter's sums are zero and what is left is the plain copy x >> 2 of the reference's other
     branch */
  XsLv c0, c1;
  c0.fill(0);
  c1.fill(0);
  XS_LANES(hb, 0, 64) {
    if (src.own(hb) >= 0) {
      int bi = 0;
      while (bi < XAAC_SBR_MAX_PATCHES - 1 && bi < XAAC_SBR_MAX_NOISE_VALUES && hb >= h->bw_borders[bi]) bi++;
      int16_t bw = (int16_t)(w->bw_array[bi] >> 16);
      const int16_t a0r = xs_mult16_shl_sat(bw, xs_m(a01.own(hb))), a0i = xs_mult16_shl_sat(bw, xs_e(a01.own(hb)));
      bw = xs_mult16_shl_sat(bw, bw);
      const int16_t a1r = xs_mult16_shl_sat(bw, xs_m(a23.own(hb))), a1i = xs_mult16_shl_sat(bw, xs_e(a23.own(hb)));
      if (bw > 0) {
        c0.own(hb) = xs_me(a0r, a0i);
        c1.own(hb) = xs_me(a1r, a1i);
      }
    }
  }
  /* the source column arrives row by row; p1 / p2 are the two rows before, which patching never rewrites (sources are
     low bands, targets high bands) */
  XsLv p2r, p2i, p1r, p1i;
  p2r.fill(0);
  p2i.fill(0);
  p1r.fill(0);
  p1i.fill(0);
  XS_UNROLL
  for (int r = 0; r < XS_REG_ROWS; r++) {
    const int l = r - 2;
    if (l + 2 >= start_idx && l < stop_idx) { /* rows start_idx - 2 .. stop_idx - 1 */
      XS_BR();
      XS_PIN(x.re[r].own(0));
      XS_PIN(x.im[r].own(0));
      const XsLv cr = x.re[r].gather(src), ci = x.im[r].gather(src);
      if (l >= start_idx) {
        XS_BR();
        XS_LANES(hb, 0, 64) {
          if (src.own(hb) >= 0) {
            const int16_t a0r = xs_m(c0.own(hb)), a0i = xs_e(c0.own(hb)), a1r = xs_m(c1.own(hb)), a1i = xs_e(c1.own(hb));
            const int32_t q1r = p1r.own(hb), q1i = p1i.own(hb), q2r = p2r.own(hb), q2i = p2i.own(hb);
            int32_t acc = fx_sub(fx_add(fx_sub(fx_mul32x16(q1r, a0r), fx_mul32x16(q1i, a0i)), fx_mul32x16(q2r, a1r)),
                                 fx_mul32x16(q2i, a1i));
            x.re[r].own(hb) = fx_add(cr.own(hb) >> 2, fx_shlw(acc, 1));
            acc = fx_add(fx_add_sat(fx_add_sat(fx_mul32x16(q1r, a0i), fx_mul32x16(q1i, a0r)), fx_mul32x16(q2r, a1i)),
                         fx_mul32x16(q2i, a1r));
            x.im[r].own(hb) = fx_add(ci.own(hb) >> 2, fx_shlw(acc, 1));
          }
        }
      }
      p2r = p1r;
      p2i = p1i;
      p1r = cr;
      p1i = ci;
    }
  }
  cx.sync();
  XS_T(14);
  XS_PAR(i, 0, h->num_if_bands) st->bw_array_prev[i] = w->bw_array[i];
  cx.sync();
}

/* env_calc.c:1211 on the register matrix: lane = band computes its own estimate, then the estimates move to the
   envelope adjuster's indexing (element c = band b0 + c) */
FX_HD void xs_energy_per_subband(const XsCx &cx, const XsQmfReg &x, int s0, int s1, int b0, int b1, int frame_exp,
                                 XsLv &est) {
  const int16_t inv_width = xaac_sbr_inv_int_table[s1 - s0];
  const int frame_exp2 = frame_exp << 1;
  XsLv eb;
  eb.fill(0);
  XS_LANES(k, 0, 64) {
    if (k >= b0 && k < b1) {
      int32_t mx = 1;
      XS_SLOTS(l) {
        if (l >= s0 && l < s1) {
          XS_BR();
          int32_t v = fx_abs_nrm(x.re[l + 2].own(k));
          if (v > mx) mx = v;
          v = fx_abs_nrm(x.im[l + 2].own(k));
          if (v > mx) mx = v;
        }
      }
      const int pre = xs_pnorm32(mx) - 4;
      int32_t accu = 0;
      int shift = 16 - pre;
      XS_SLOTS(l) {
        if (l >= s0 && l < s1) {
          XS_BR();
          int16_t t = shift > 0 ? (int16_t)xs_sar(x.re[l + 2].own(k), shift) : (int16_t)xs_shl(x.re[l + 2].own(k), -shift);
          accu = fx_add(accu, (int32_t)t * t);
          t = shift > 0 ? (int16_t)xs_sar(x.im[l + 2].own(k), shift) : (int16_t)xs_shl(x.im[l + 2].own(k), -shift);
          accu = fx_add(accu, (int32_t)t * t);
        }
      }
      int32_t e = 0;
      if (accu != 0) {
        shift = -xs_pnorm32(accu);
        int16_t sum_m = (int16_t)xs_shr_dir_sat_limit(accu, 16 + shift);
        sum_m = xs_mult16_shl_sat(sum_m, inv_width);
        shift = shift - (pre << 1);
        e = xs_me(sum_m, (int16_t)(frame_exp2 + shift + 1));
      }
      eb.own(k) = e;
    }
  }
  const XsLv ec = eb.shifted(cx, b0);
  XS_LANES(c, 0, b1 - b0) est.own(c) = ec.own(c);
}

/* env_calc.c:1298 (interpol_freq == 0) is not restated for the register matrix: such a frame goes through the
   matrix-in-memory form (the kernel's list launch; the oracle's own path) */
FX_HD void xs_energy_per_sfb(const XsCx &, const XsQmfReg &, int, const int16_t *, int, int, int, int, XsWork *, XsLv &) {}
FX_HD bool xs_reg_core_takes(const xaac_sbr_header *h) { return h->interpol_freq != 0; }

/* env_calc.c:479 (HQ branch) on the register matrix; see xs_adapt_noise_gain_hq in sbr_core.h for the algorithm.  Lane b
   owns band b: filter-buffer entry i = b - (sub-band start) and, from max_qmf_subband_aac (the `sb_start` argument, as
   the caller passes it) on, adjusted band k = b - sb_start. */
template <class ST>
FX_HD void xs_adapt_noise_gain_hq(const XsCx &cx, ST *st, XsEnv &v, int noise_e, int nsb, int skip, int s0, int s1,
                                  int input_e, int adj_e, int final_e, int sb_start, int noise_absc,
                                  int smooth_length, const XsQmfReg &x) {
  const int bands = nsb - skip;
  const int start_up = cx.uni(st->start_up);
  const int ph0 = cx.uni(st->ph_index), harm0 = cx.uni(st->harm_index);
  const int fb_noise_e0 = start_up ? noise_e : cx.uni(st->filt_buf_noise_e);
  cx.sync();
  XS_LANES(k, 0, bands) {
    int16_t g[2] = {xs_m(v.gain.own(k)), xs_e(v.gain.own(k))};
    if (start_up) {
      st->filt_buf_me[2 * (skip + k)] = g[0];
      st->filt_buf_me[2 * (skip + k) + 1] = g[1];
      st->filt_buf_noise_m[skip + k] = xs_m(v.noise.own(k));
    } else {
      xs_equalize_filt_buf(&st->filt_buf_me[2 * (skip + k)], g);
      v.gain.own(k) = xs_me(g[0], g[1]);
    }
  }
  cx.sync();
  /* band values seen from the lane that owns the band */
  const XsLv gain_b = v.gain.shifted(cx, -sb_start), noise_b = v.noise.shifted(cx, -sb_start),
             sine_b = v.sine.shifted(cx, -sb_start);
  XsLv noise_out;
  noise_out.fill(0);
  XS_T(21);
  const int first_i = sb_start - skip; /* band of filter-buffer entry 0 */
  const int n_smooth = s1 - s0 < smooth_length ? (s1 > s0 ? s1 - s0 : 0) : smooth_length;
  const bool crosses = s0 < 32 && s1 > 32; /* the envelope runs over the frame's slot 32: the scale changes there */
  XS_LANES(b, 0, 64) {
    const int i = b - first_i;
    if (i >= 0 && i < nsb) {
      const int k = i - skip;
      const int16_t gm = xs_m(gain_b.own(b)), ge = xs_e(gain_b.own(b));
      const int16_t sm = xs_m(sine_b.own(b)), se = xs_e(sine_b.own(b));
      const int16_t nl0 = xs_m(noise_b.own(b));
      int16_t fbm = st->filt_buf_me[2 * i], fbn = st->filt_buf_noise_m[i];
      const int kk = k >= 0 ? k : 0;
      const bool live = k >= 0;
      /* the per-slot quantities that only depend on which side of slot 32 the slot lies: A below, B from 32 on */
      const int ne_a = noise_e, ne_b = s0 < 32 ? final_e : noise_e;
      const int16_t nl_a = nl0, nl_b = (s0 < 32 && live) ? xs_noise_rescale(nl0, final_e - noise_e) : nl0;
      int ls_a, rs_a, keep_a, ls_b, rs_b, keep_b;
      {
        const int shift = (int16_t)(ge - (int16_t)((int16_t)(adj_e - input_e) - 1));
        const int m = (shift > 0 ? shift : -shift) & 0xff;
        ls_a = shift > 0 && m <= 31 ? m : 0;
        rs_a = shift > 0 ? 0 : (m < 31 ? m : 31);
        keep_a = shift > 0 && m > 31 ? 0 : -1;
      }
      {
        const int shift = (int16_t)(ge - (int16_t)((int16_t)(final_e - input_e) - 1));
        const int m = (shift > 0 ? shift : -shift) & 0xff;
        ls_b = shift > 0 && m <= 31 ? m : 0;
        rs_b = shift > 0 ? 0 : (m < 31 ? m : 31);
        keep_b = shift > 0 && m > 31 ? 0 : -1;
      }
      const int tmp_a = (int16_t)(se - (int16_t)(ne_a - 16)), tmp_b = (int16_t)(se - (int16_t)(ne_b - 16));
      const int32_t sle_a = tmp_a > 0 ? fx_shl(sm, tmp_a) : fx_shr(sm, tmp_a); /* (sic) env_calc.c:1797 */
      const int32_t slo_a = tmp_a > 0 ? fx_shl(sm, tmp_a) : fx_shr(sm, -tmp_a);
      const int32_t sle_b = tmp_b > 0 ? fx_shl(sm, tmp_b) : fx_shr(sm, tmp_b);
      const int32_t slo_b = tmp_b > 0 ? fx_shl(sm, tmp_b) : fx_shr(sm, -tmp_b);
      const bool tone = live && sm != 0, noise = live && sm == 0 && !noise_absc;
      const bool fi = ((sb_start ^ kk) & 1) != 0;
      /* the filter buffer's noise follows the noise exponent: into the envelope's first slot, and across slot 32 */
      if (s1 > s0) fbn = xs_noise_rescale(fbn, fb_noise_e0 - noise_e);
      XS_SLOTS(l) {
        if (l >= s0 && l < s1) {
          XS_BR();
          const bool up = l >= 32; /* a constant of the unrolled body */
          if (l == 32 && s0 < 32) fbn = xs_noise_rescale(fbn, ne_a - ne_b);
          const int j = l - s0;
          int16_t sg = gm, snz = up ? nl_b : nl_a;
          if (j < n_smooth && live) { /* ixheaacd_adj_timeslot's smoothed start (env_dec.c:845): at most four slots */
            const int16_t smooth = xaac_sbr_smooth_filter[j & 3];
            if (smooth) {
              const int16_t direct = fx_sat16(0x7fff - (int32_t)smooth);
              const int16_t t = (int16_t)(xs_mult16(smooth, fbm) + xs_mult16(direct, gm));
              const int16_t t1 = (int16_t)(xs_mult16(smooth, fbn) + xs_mult16(direct, snz));
              fbm = (int16_t)(t << 1);
              fbn = (int16_t)(t1 << 1);
              sg = fbm;
              snz = fbn;
            }
          }
          if (live) {
            const int ph = (ph0 + j * bands) & 511, hi = (harm0 + j) & 3;
            const int32_t rp = XS_TAB_RAND(ph + 1 + kk);
            int32_t re = fx_mul32x16(x.re[l + 2].own(b), sg), im = fx_mul32x16(x.im[l + 2].own(b), sg);
            re = (fx_shlw(re, up ? ls_b : ls_a) >> (up ? rs_b : rs_a)) & (up ? keep_b : keep_a);
            im = (fx_shlw(im, up ? ls_b : ls_a) >> (up ? rs_b : rs_a)) & (up ? keep_b : keep_a);
            const int32_t sl_even = up ? sle_b : sle_a, sl_odd = up ? slo_b : slo_a;
            const bool plus = fi != (hi == 1);
            const int32_t re_t = hi == 0 ? fx_add_sat(re, sl_even) : (hi == 2 ? fx_sub_sat(re, sl_even) : re);
            const int32_t im_t = (hi & 1) ? (plus ? fx_add_sat(im, sl_odd) : fx_sub_sat(im, sl_odd)) : im;
            const int32_t re_n = xs_mac16x16_shl_sat(re, (int16_t)(rp >> 16), snz);
            const int32_t im_n = xs_mac16x16_shl_sat(im, (int16_t)rp, snz);
            x.re[l + 2].own(b) = tone ? re_t : (noise ? re_n : re);
            x.im[l + 2].own(b) = tone ? im_t : (noise ? im_n : im);
          }
        }
      }
      st->filt_buf_me[2 * i] = fbm;
      st->filt_buf_noise_m[i] = fbn;
      noise_out.own(b) = crosses ? nl_b : nl_a;
    }
  }
  cx.sync();
  XS_T(22);
  {
    const XsLv nb = noise_out.shifted(cx, sb_start); /* back to the adjuster's indexing */
    XS_LANES(k, 0, bands) {
      v.noise.own(k) = xs_me((int16_t)nb.own(k), xs_e(v.noise.own(k)));
      st->filt_buf_me[2 * (skip + k)] = xs_m(v.gain.own(k)); /* env_calc.c:1060 */
      st->filt_buf_noise_m[skip + k] = (int16_t)nb.own(k);
    }
  }
  XS_ONE {
    const int n = s1 > s0 ? s1 - s0 : 0;
    int ne = noise_e;
    if (s0 < 32 && s1 > 32) ne = final_e;
    st->start_up = 0;
    st->filt_buf_noise_e = n > 0 ? ne : fb_noise_e0;
    st->ph_index = (int16_t)((ph0 + n * bands) & 511);
    st->harm_index = (int16_t)((harm0 + n) & 3);
  }
  cx.sync();
}

#endif /* XAAC_SBR_CORE_REG_H */
