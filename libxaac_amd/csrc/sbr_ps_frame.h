/*
 * sbr_ps_frame.h -- one whole frame (32 QMF slots) of the fixed-point parametric-stereo tool, arranged for a
 * 64-lane wave instead of for a slot loop.
 *
 * The reference runs the tool slot by slot inside the left channel's synthesis loop (decoder/ixheaacd_qmf_dec.c:
 * 1015-1031 -> ixheaacd_apply_ps, thumb_ps_dec.c:69).  Of everything it does per slot only three things are
 * recursions over the slots:
 *   - the transient detector's three smoothed values per bin            (ps_dec.c:547-590),
 *   - the all-pass chains of the 10 hybrid sub-bands and QMF bands 3..22 (ps_dec.c:236 / :339: delay lines),
 *   - the envelope counter that decides where new mixing coefficients start (qmf_dec.c:1019).
 * The rest -- hybrid analysis (a FIR), band powers, the bin sums, the transient ratio's division, the plain 14- and
 * 1-slot delays, the interpolated 2x2 rotation (H advances by a constant per slot: H_l = H_0 + n * delta mod 2^16),
 * the hybrid synthesis sums, the scale shifts -- depends on its slot only.  So the frame runs in phases:
 *
 *   P1 hybrid analysis of all 32 slots            lane = slot / (band, slot)
 *   P2 envelope walk: segments of constant delta  scalar; coefficients of each border: lane = parameter group
 *   P3 band powers, all-pass / delay inputs       lane = QMF band, loop over slots (coalesced row reads)
 *      group sums of the upper bins               lane = (slot, group): the addends are >= 0, so the saturating
 *                                                 sum is min(MAX, exact sum) whatever the order
 *   P4 transient detector                         lane = bin, loop over slots (the recursion), then the 640 ratios
 *                                                 (one division each) lane-parallel
 *   P5 all-pass chains                            lane = chain (30), loop over slots (the recursion)
 *   P6 rotation in the hybrid domain + its sums   lane = (slot, sub-band)
 *   P7 delays, rotation, output scaling           lane = QMF band, loop over slots (coalesced row reads / writes)
 * Every value is computed by the same operations in the same order as in the slot loop (sbr_ps.h, which stays the
 * oracle's restatement of the reference); what changes is only when.  The same source compiled for the host with
 * lane count 1 is checked against that slot loop on the reference's captured frames and on fuzzed side info
 * (tests/test_ps_frame_cpu.py) before the GPU sees it.
 *
 * Borders no parser produces are handled as the slot loop would: the envelope counter only ever looks at its
 * current border (a border that lies behind the current slot is never reached), and if the first border is not
 * slot 0 the band limit `usb` (and the clearing of newly active all-pass delay lines, ps_dec.c:733-757) switches
 * at that slot.
 */
#ifndef XAAC_SBR_PS_FRAME_H
#define XAAC_SBR_PS_FRAME_H

#include "sbr_ps.h"

#ifndef XP_T
#define XP_T(i) /* optional phase timer hook (tools/prof_sbr_core.py ps) */
#endif

#define XP_MAX_SEG (XAAC_PS_MAX_ENV + 2) /* the segment carried in from the last frame + one per border (env 0..5) */

struct XpFrameWork {
  union {                    /* three scratch areas that are never live together */
    int32_t hyb_u[3][2][44]; /* P1: hybrid filter input of QMF bands 0..2: 12 slots of history + this frame's 32 */
    int32_t gsum[8][56];     /* P3: addends of the group sums, bands 9..63 of eight slots */
    int32_t low[32][12];     /* P6 -> P7: QMF bands 0..2 after hybrid synthesis: [band][l_re, l_im, r_re, r_im] */
  };
  int32_t hyb_l[32][20];     /* left hybrid sub-band samples of every slot: re 0..9 | im 10..19; rotated in place */
  union {
    int32_t binpw[32][20];   /* P3: bin powers; P4: smoothed energy ... */
    int16_t ratio[32][20];   /* ... compacted in place into the transient ratios (entry i lands inside entry i / 2) */
  };
  union {
    int32_t peak[32][20];    /* P4: transient peak difference */
    int32_t hyb_r[32][20];   /* P6: right hybrid sub-band samples after the rotation */
  };
  uint32_t ap[32][30];       /* all-pass chains, 10 hybrid + QMF bands 3..22: rounded input pairs -> output pairs */
  uint32_t dl[32][12];       /* rounded input pairs of QMF bands 23..34 (the 14-slot delay) */
  int16_t seg_h[XP_MAX_SEG][4][24]; /* per segment and group: H11, H12, H21, H22 before the segment's first slot */
  int16_t seg_d[XP_MAX_SEG][4][24]; /* per-slot increments */
  int8_t seg_of_slot[32];
  int8_t seg_start[8];
};

FX_HD uint32_t xp_pack16(int16_t lo, int16_t hi) { return (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16); }
FX_HD int16_t xp_lo16(uint32_t v) { return (int16_t)(v & 0xffffu); }
FX_HD int16_t xp_hi16(uint32_t v) { return (int16_t)(v >> 16); }

FX_HD int32_t xp_adj_word(int32_t v, int shift) { /* env_calc.c:1099 on one word */
  if (shift == 0) return v;
  if (shift > 31) shift = 31;
  if (shift < -31) shift = -31;
  return shift > 0 ? fx_shlw(v, shift) : (v >> -shift);
}

/* ps_dec.c:470-519: the eight transient-detector bins that live in the hybrid domain */
FX_HD int32_t xp_bin_power_hyb(const XpTables *T, int bin, const int32_t *re, const int32_t *im) {
  if (bin < 2) {
    const int a = bin == 0 ? 0 : 4, b = bin == 0 ? 5 : 1;
    int32_t pw = xp_power(re[a], im[a]);
    pw = fx_add_sat(pw, fx_mul32x16(re[b], (int16_t)(re[b] >> 16)));
    return fx_add_sat(pw, fx_mul32x16(im[b], (int16_t)(im[b] >> 16)));
  }
  const int sb = T->borders_group[bin + 2];
  return xp_power(re[sb], im[sb]);
}

/* One frame.  xl: the stream's QMF matrix, slot 0 at xl (rows of 64 re | 64 im; rows 0..37 are read, rows 0..31
   are rewritten with the left channel in the scale the synthesis bank expects); xr: 32 rows out, the right channel.
   lb/ov_lb/hb_scale, st_syn, lsb, usb: what the SBR core left for the synthesis bank.  Returns ps_scale. */
template <class PS>
FX_HD int xp_ps_frame(const XsCx &cx, const XpTables *T, PS *ps, const xaac_ps_frame *pf, XpFrameWork *w, int32_t *xl,
                      int32_t *xr, int lb_scale, int ov_lb_scale, int hb_scale, int st_syn, int lsb, int usb) {
  const int ps_scale = xp_init_ps_scale(cx, ps, lb_scale, ov_lb_scale, hb_scale); /* sbr_dec.c:1252 */
  const int ov_lb_shift = ps_scale - ov_lb_scale, lb_shift = ps_scale - lb_scale, hb_shift = ps_scale - hb_scale;
  const int common_shift = (st_syn - ps_scale) - 8;
  XP_T(1);

  /* ---- P1: hybrid analysis (hybrid.c:214: a 13-tap FIR on QMF bands 0..2 looking six slots ahead).  Input of step
     l: row l + 6 as adjust_scale leaves it (qmf_dec.c:937: slots of the next frame are not rescaled), then the
     delay-buffer shift of thumb_ps_dec.c:77. */
  XS_PAR(l, 0, 32) {
    const int shiftdelay = l < 32 - 6 ? 0 : (int16_t)(lb_scale - ps_scale);
    for (int b = 0; b < 3; b++) {
      const int sha = l + 6 < 32 ? (b < lsb ? lb_shift : (b < usb ? hb_shift : 0)) : 0;
      for (int c = 0; c < 2; c++) {
        int32_t v = xp_adj_word(xl[(l + 6) * 128 + 64 * c + b], sha);
        v = shiftdelay < 0 ? fx_shl(v, -shiftdelay) : fx_shr(v, shiftdelay);
        w->hyb_u[b][c][12 + l] = v;
      }
    }
  }
  XS_PAR(i, 0, 12)
    for (int b = 0; b < 3; b++) {
      w->hyb_u[b][0][i] = ps->hyb_buf[b][0][i];
      w->hyb_u[b][1][i] = ps->hyb_buf[b][1][i];
    }
  cx.sync();
  XS_PAR(l, 0, 32) { /* QMF band 0: eight-channel filter, six sub-bands */
    int32_t re[8], im[8];
    xp_filt_8ch(T, &w->hyb_u[0][0][l], &w->hyb_u[0][1][l], re, im);
    for (int k = 0; k < 6; k++) {
      w->hyb_l[l][k] = re[k];
      w->hyb_l[l][10 + k] = im[k];
    }
  }
  XS_PAR(i, 0, 64) { /* QMF bands 1 and 2: two sub-bands each */
    const int b = 1 + (i >> 5), l = i & 31;
    int32_t re[2], im[2];
    xp_filt_2ch(T, &w->hyb_u[b][0][l], &w->hyb_u[b][1][l], re, im);
    w->hyb_l[l][4 + 2 * b] = re[0];
    w->hyb_l[l][5 + 2 * b] = re[1];
    w->hyb_l[l][14 + 2 * b] = im[0];
    w->hyb_l[l][15 + 2 * b] = im[1];
  }
  XS_PAR(i, 0, 12)
    for (int b = 0; b < 3; b++) {
      ps->hyb_buf[b][0][i] = w->hyb_u[b][0][32 + i];
      ps->hyb_buf[b][1][i] = w->hyb_u[b][1][32 + i];
    }
  cx.sync();
  XP_T(2);

  /* ---- P2: the envelope walk of qmf_dec.c:1019 ("if slot == border[env]: init_rot_env; env++").  Segment 0
     continues the last frame's interpolation from the state; every border reached starts a new one. */
  XS_PAR(g, 0, XAAC_PS_GROUPS) {
    w->seg_h[0][0][g] = ps->H11_H12[2 * g];
    w->seg_h[0][1][g] = ps->H11_H12[2 * g + 1];
    w->seg_h[0][2][g] = ps->H21_H22[2 * g];
    w->seg_h[0][3][g] = ps->H21_H22[2 * g + 1];
    w->seg_d[0][0][g] = ps->delta_h11_h12[2 * g];
    w->seg_d[0][1][g] = ps->delta_h11_h12[2 * g + 1];
    w->seg_d[0][2][g] = ps->delta_h21_h22[2 * g];
    w->seg_d[0][3][g] = ps->delta_h21_h22[2 * g + 1];
  }
  XS_ONE w->seg_start[0] = 0;
  const int usb_prev = cx.uni(ps->usb);
  int clear_slot = 32; /* first slot that runs with the new usb: the slot of border 0, if it is reached at all */
  {
    int env = 0, cur = 0, nseg = 1;
    for (int l = 0; l < 32; l++) {
      if (env <= XAAC_PS_MAX_ENV && l == cx.uni(pf->border_position[env])) {
        if (env == 0) clear_slot = l;
        cx.sync();
        xp_rot_env_coeffs(cx, T, ps, pf, env); /* ps->H.. = the old targets, ps->delta.., ps->h.._vec = the new ones */
        XS_PAR(g, 0, XAAC_PS_GROUPS) {
          w->seg_h[nseg][0][g] = ps->H11_H12[2 * g];
          w->seg_h[nseg][1][g] = ps->H11_H12[2 * g + 1];
          w->seg_h[nseg][2][g] = ps->H21_H22[2 * g];
          w->seg_h[nseg][3][g] = ps->H21_H22[2 * g + 1];
          w->seg_d[nseg][0][g] = ps->delta_h11_h12[2 * g];
          w->seg_d[nseg][1][g] = ps->delta_h11_h12[2 * g + 1];
          w->seg_d[nseg][2][g] = ps->delta_h21_h22[2 * g];
          w->seg_d[nseg][3][g] = ps->delta_h21_h22[2 * g + 1];
        }
        XS_ONE w->seg_start[nseg] = (int8_t)l;
        cur = nseg++;
        env++;
      }
      XS_ONE w->seg_of_slot[l] = (int8_t)cur;
    }
  }
  cx.sync();
  XP_T(3);

  /* ---- P3: band powers (ps_dec.c:520-545), the inputs of the all-pass chains and of the 14-slot delay.  A lane
     walks its band through eight slots at a time: the sixteen row words are fetched together. */
  for (int c = 0; c < 4; c++) {
    XS_PAR(sb, 0, 64) {
      int32_t rre[8], rim[8];
      for (int ls = 0; ls < 8; ls++) {
        rre[ls] = xl[(8 * c + ls) * 128 + sb];
        rim[ls] = xl[(8 * c + ls) * 128 + 64 + sb];
      }
      const int gsh = sb < 11 ? 0 : (sb < 18 ? 1 : (sb < 23 ? 2 : (sb < 35 ? 3 : 4))); /* group_shift of the band's group */
      for (int ls = 0; ls < 8; ls++) {
        const int l = 8 * c + ls;
        const int usb_l = l >= clear_slot ? usb : usb_prev;
        const int sh = sb < lsb ? (l < 6 ? ov_lb_shift : lb_shift) : (sb < usb ? hb_shift : 0);
        const int32_t re = xp_adj_word(rre[ls], sh), im = xp_adj_word(rim[ls], sh);
        if (sb >= 3) {
          const int32_t pw = xp_power(re, im);
          if (sb < 23)
            w->ap[l][10 + sb - 3] = xp_pack16(fx_round16(re), fx_round16(im));
          else if (sb < 35)
            w->dl[l][sb - 23] = xp_pack16(fx_round16(re), fx_round16(im));
          if (sb < 9)
            w->binpw[l][sb + 5] = pw;
          else
            w->gsum[ls][sb - 9] = sb < usb_l ? (pw >> gsh) : 0;
        }
      }
    }
    cx.sync();
    XS_PAR(i, 0, 48) { /* bins 14..19 = sums over the groups [9,11) [11,14) [14,18) [18,23) [23,35) [35,64) */
      const int ls = i / 6, g = i % 6;
      const int b0 = T->borders_group[16 + g], b1 = T->borders_group[17 + g];
      int32_t acc = 0;
      for (int sb = b0; sb < b1; sb++) acc = fx_add_sat(acc, w->gsum[ls][sb - 9]);
      w->binpw[8 * c + ls][14 + g] = acc;
    }
    cx.sync();
  }
  XS_PAR(i, 0, 320) {
    const int l = i / 10, u = i % 10;
    w->ap[l][u] = xp_pack16(fx_round16(w->hyb_l[l][u]), fx_round16(w->hyb_l[l][10 + u]));
  }
  XS_PAR(i, 0, 256) {
    const int l = i >> 3, bin = i & 7;
    w->binpw[l][bin] = xp_bin_power_hyb(T, bin, &w->hyb_l[l][0], &w->hyb_l[l][10]);
  }
  cx.sync();
  XP_T(4);

  /* ---- P4: transient detector (ps_dec.c:547-590): peak decay against smoothed energy, per bin */
  XS_PAR(bin, 0, 20) {
    int32_t pd = ps->peak_decay_diff[bin], pdp = ps->peak_decay_diff_prev[bin], nrg = ps->energy_prev[bin];
    for (int l = 0; l < 32; l++) {
      int32_t pw = fx_shl(w->binpw[l][bin], 1);
      if (pw < 0) pw = 0;
      pd = fx_mul32x16_shl(pd, 0x620a);
      if (pw > pd) pd = pw;
      pdp = fx_add_sat(fx_mul32x16_shl(pdp, 0x6000), fx_sub_sat(pd, pw) >> 2);
      nrg = fx_add_sat(fx_mul32x16_shl(nrg, 0x6000), pw >> 2);
      w->binpw[l][bin] = nrg;
      w->peak[l][bin] = fx_add_sat(pdp, pdp >> 1);
    }
    ps->peak_decay_diff[bin] = pd;
    ps->peak_decay_diff_prev[bin] = pdp;
    ps->energy_prev[bin] = nrg;
  }
  cx.sync();
  for (int i0 = 0; i0 < 640; i0 += 64) { /* every lane reads its (energy, peak) pair before any lane stores a ratio */
    int16_t q = 0;
    XS_PAR(i, i0, i0 + 64) {
      const int32_t pk = (&w->peak[0][0])[i], nrg = (&w->binpw[0][0])[i];
      q = pk <= nrg ? (int16_t)0x7fff : (int16_t)xp_divide16_pos(nrg, pk);
#if !defined(__HIP_DEVICE_COMPILE__)
      (&w->ratio[0][0])[i] = q; /* sequential: entry i / 2 has been consumed */
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__)
    cx.sync();
    (&w->ratio[0][0])[i0 + cx.lane] = q;
    cx.sync();
#endif
  }
  cx.sync();
  XP_T(5);

  /* ---- P5: the thirty all-pass chains (ps_dec.c:236 hybrid sub-bands, :339 QMF bands 3..22 whatever usb is) */
  {
    int idx = cx.uni(ps->idx);
    int is0 = cx.uni(ps->idx_ser[0]), is1 = cx.uni(ps->idx_ser[1]), is2 = cx.uni(ps->idx_ser[2]);
    const int ss0 = cx.uni(ps->sample_ser[0]), ss1 = cx.uni(ps->sample_ser[1]), ss2 = cx.uni(ps->sample_ser[2]);
    for (int l = 0; l < 32; l++) {
      if (l == clear_slot && usb > usb_prev && usb_prev) { /* ps_dec.c:733-757: bands that just became active */
        const int ap_hi = usb < 23 ? usb : 23;
        cx.sync();
        if (ap_hi > usb_prev)
          for (int i = 0; i < 3; i++)
            for (int j = 0; j < (i == 0 ? ss0 : (i == 1 ? ss1 : ss2)); j++)
              XS_PAR(k, 2 * usb_prev, 2 * ap_hi) ps->ser[j][i][k] = 0;
        cx.sync();
      }
      XS_PAR(u, 0, 30) {
        const int hyb = u < 10, sb = hyb ? u : u - 7;
        const int di = 9 + 3 * (sb - 3);
        int16_t *d0 = hyb ? &ps->sub[idx][2 * sb] : &ps->ap[idx][2 * sb];
        int16_t *e0 = hyb ? &ps->sub_ser[is0][0][2 * sb] : &ps->ser[is0][0][2 * sb];
        int16_t *e1 = hyb ? &ps->sub_ser[is1][1][2 * sb] : &ps->ser[is1][1][2 * sb];
        int16_t *e2 = hyb ? &ps->sub_ser[is2][2][2 * sb] : &ps->ser[is2][2][2 * sb];
        const int16_t *ph = hyb ? &T->frac_delay_phase_fac_qmf_sub_re_im[2 * sb] : &T->frac_delay_phase_fac_qmf_re_im[2 * sb];
        const int16_t *pser = hyb ? &T->frac_delay_phase_fac_qmf_sub_ser_re_im[2 * sb] : &T->frac_delay_phase_fac_qmf_ser_re_im[2 * sb];
        const int pstep = hyb ? 32 : 64;
        const uint32_t in = w->ap[l][u];
        int16_t o_re, o_im;
        xp_allpass(d0, xp_lo16(in), xp_hi16(in), ph, e0, e1, e2, pser, pser + pstep, pser + 2 * pstep,
                   hyb ? T->rev_link_decay_ser[0] : T->decay_scale_factor[di],
                   hyb ? T->rev_link_decay_ser[1] : T->decay_scale_factor[di + 1],
                   hyb ? T->rev_link_decay_ser[2] : T->decay_scale_factor[di + 2], &o_re, &o_im);
        w->ap[l][u] = xp_pack16(o_re, o_im);
      }
      idx = idx + 1 >= 2 ? 0 : idx + 1;
      is0 = is0 + 1 >= ss0 ? 0 : is0 + 1;
      is1 = is1 + 1 >= ss1 ? 0 : is1 + 1;
      is2 = is2 + 1 >= ss2 ? 0 : is2 + 1;
    }
    cx.sync();
    XS_ONE {
      ps->idx = (int16_t)idx;
      ps->idx_ser[0] = (int16_t)is0;
      ps->idx_ser[1] = (int16_t)is1;
      ps->idx_ser[2] = (int16_t)is2;
    }
  }
  cx.sync();
  XP_T(6);

  /* ---- P6: rotation of the hybrid sub-bands (ps_dec.c:856, groups 0..9) and hybrid synthesis of QMF bands 0..2 */
  XS_PAR(i, 0, 320) {
    const int l = i / 10, sb = i % 10;
    const int s = w->seg_of_slot[l], n = l - w->seg_start[s] + 1;
    const int16_t h11 = (int16_t)(w->seg_h[s][0][sb] + n * w->seg_d[s][0][sb]);
    const int16_t h12 = (int16_t)(w->seg_h[s][1][sb] + n * w->seg_d[s][1][sb]);
    const int16_t h21 = (int16_t)(w->seg_h[s][2][sb] + n * w->seg_d[s][2][sb]);
    const int16_t h22 = (int16_t)(w->seg_h[s][3][sb] + n * w->seg_d[s][3][sb]);
    const int16_t tr = w->ratio[l][T->hybrid_to_bin[sb]];
    const uint32_t o = w->ap[l][sb];
    int32_t l_re = w->hyb_l[l][sb], l_im = w->hyb_l[l][10 + sb];
    int32_t r_re = xp_m16x16_shl(xp_lo16(o), tr), r_im = xp_m16x16_shl(xp_hi16(o), tr);
    xp_rotate(&l_re, &r_re, h11, h12, h21, h22);
    xp_rotate(&l_im, &r_im, h11, h12, h21, h22);
    w->hyb_l[l][sb] = l_re;
    w->hyb_l[l][10 + sb] = l_im;
    w->hyb_r[l][sb] = r_re;
    w->hyb_r[l][10 + sb] = r_im;
  }
  cx.sync();
  XS_PAR(i, 0, 384) {
    const int l = i / 12, b = (i % 12) >> 2, c = i & 3;
    const int p = b == 0 ? 0 : 4 + 2 * b, n = b == 0 ? 6 : 2;
    const int32_t *src = (c < 2 ? &w->hyb_l[l][0] : &w->hyb_r[l][0]) + ((c & 1) ? 10 : 0) + p;
    int32_t a = src[0];
    for (int k = 1; k < n; k++) a = fx_add_sat(a, src[k]);
    w->low[l][4 * b + c] = a;
  }
  cx.sync();
  XP_T(7);

  /* ---- P7: plain delays (ps_dec.c:602-648), rotation (ps_dec.c:893-945), the common shift in front of the left
     bank (generic:1610).  A lane walks its band through the 32 slots (one row word pair in, two out per slot, eight
     slots' words fetched together); what the slot loop keeps in the state between slots -- the interpolated
     coefficients of the band's group, the 1-slot delay -- stays in the lane's registers. */
  {
    const int idx_long0 = cx.uni(ps->idx_long);
    XS_PAR(sb, 0, 64) {
      const int g = T->band_to_group[sb];
      int16_t h11 = 0, h12 = 0, h21 = 0, h22 = 0, d11 = 0, d12 = 0, d21 = 0, d22 = 0;
      uint32_t prev = sb >= 35 ? xp_pack16(ps->sd[2 * (sb - 35)], ps->sd[2 * (sb - 35) + 1]) : 0u;
      for (int c = 0; c < 4; c++) {
        int32_t rre[8], rim[8];
        for (int ls = 0; ls < 8; ls++) {
          rre[ls] = xl[(8 * c + ls) * 128 + sb];
          rim[ls] = xl[(8 * c + ls) * 128 + 64 + sb];
        }
        for (int ls = 0; ls < 8; ls++) {
          const int l = 8 * c + ls;
          const int s = cx.uni(w->seg_of_slot[l]);
          const int usb_l = l >= clear_slot ? usb : usb_prev;
          if (l == cx.uni(w->seg_start[s])) { /* a border: this group's coefficients restart from the old targets */
            h11 = w->seg_h[s][0][g]; h12 = w->seg_h[s][1][g]; h21 = w->seg_h[s][2][g]; h22 = w->seg_h[s][3][g];
            d11 = w->seg_d[s][0][g]; d12 = w->seg_d[s][1][g]; d21 = w->seg_d[s][2][g]; d22 = w->seg_d[s][3][g];
          }
          h11 = (int16_t)(h11 + d11); /* the interpolation advances whether or not the band is rotated */
          h12 = (int16_t)(h12 + d12);
          h21 = (int16_t)(h21 + d21);
          h22 = (int16_t)(h22 + d22);
          const int sh = sb < lsb ? (l < 6 ? ov_lb_shift : lb_shift) : (sb < usb ? hb_shift : 0);
          int32_t re = xp_adj_word(rre[ls], sh), im = xp_adj_word(rim[ls], sh);
          int32_t r_re = 0, r_im = 0;
          if (sb < 3) {
            re = w->low[l][4 * sb];
            im = w->low[l][4 * sb + 1];
            r_re = w->low[l][4 * sb + 2];
            r_im = w->low[l][4 * sb + 3];
          } else if (sb < usb_l) {
            uint32_t o;
            int16_t tr;
            if (sb < 23) {
              o = w->ap[l][10 + sb - 3];
              tr = w->ratio[l][T->delay_to_bin[sb]];
            } else if (sb < 35) { /* what slot l - 14 put in, if it ran with this band active; else the state */
              const int pos = (idx_long0 + l) % 14;
              o = (l >= 14 && sb < (l - 14 >= clear_slot ? usb : usb_prev))
                      ? w->dl[l - 14][sb - 23]
                      : xp_pack16(ps->ld[pos][2 * (sb - 23)], ps->ld[pos][2 * (sb - 23) + 1]);
              tr = w->ratio[l][18];
            } else {
              o = prev;
              tr = w->ratio[l][19];
              prev = xp_pack16(fx_round16(re), fx_round16(im));
            }
            r_re = xp_m16x16_shl(xp_lo16(o), tr);
            r_im = xp_m16x16_shl(xp_hi16(o), tr);
            xp_rotate(&re, &r_re, h11, h12, h21, h22);
            xp_rotate(&im, &r_im, h11, h12, h21, h22);
          }
          if (common_shift < 0) {
            const int cs = -common_shift > 31 ? 31 : -common_shift;
            re = fx_shr(re, cs);
            im = fx_shr(im, cs);
          } else if (common_shift > 0) {
            re = fx_shl_sat(re, common_shift);
            im = fx_shl_sat(im, common_shift);
          }
          xl[l * 128 + sb] = re;
          xl[l * 128 + 64 + sb] = im;
          xr[l * 128 + sb] = r_re;
          xr[l * 128 + 64 + sb] = r_im;
        }
      }
      if (sb >= 35) { /* the 1-slot delay line as the last slot leaves it */
        ps->sd[2 * (sb - 35)] = xp_lo16(prev);
        ps->sd[2 * (sb - 35) + 1] = xp_hi16(prev);
      }
    }
    cx.sync();
    /* the delay lines as the slot loop leaves them: each position of the 14-slot ring holds the input of the last
       slot that wrote it (a slot writes band sb only while sb < usb) */
    XS_PAR(i, 0, 14 * 12) {
      const int p = i / 12, j = i % 12, sb = 23 + j;
      const int l0 = (p - idx_long0 % 14 + 14) % 14;
      for (int l = l0 + 28; l >= 0; l -= 14) {
        if (l < 32 && sb < (l >= clear_slot ? usb : usb_prev)) {
          ps->ld[p][2 * j] = xp_lo16(w->dl[l][j]);
          ps->ld[p][2 * j + 1] = xp_hi16(w->dl[l][j]);
          break;
        }
      }
    }
    const int s = cx.uni(w->seg_of_slot[31]), n = 32 - cx.uni(w->seg_start[s]);
    XS_PAR(g, 0, XAAC_PS_GROUPS) {
      ps->H11_H12[2 * g] = (int16_t)(w->seg_h[s][0][g] + n * w->seg_d[s][0][g]);
      ps->H11_H12[2 * g + 1] = (int16_t)(w->seg_h[s][1][g] + n * w->seg_d[s][1][g]);
      ps->H21_H22[2 * g] = (int16_t)(w->seg_h[s][2][g] + n * w->seg_d[s][2][g]);
      ps->H21_H22[2 * g + 1] = (int16_t)(w->seg_h[s][3][g] + n * w->seg_d[s][3][g]);
    }
    XS_ONE {
      ps->idx_long = (int16_t)((idx_long0 + 32) % 14);
      if (clear_slot < 32) ps->usb = (int16_t)usb;
    }
  }
  cx.sync();
  XP_T(8);
  return ps_scale;
}

#endif /* XAAC_SBR_PS_FRAME_H */
