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

#define XP_NO_PS_SCALE 0x7fffffff
#define XP_MAX_SEG (XAAC_PS_MAX_ENV + 2) /* the segment carried in from the last frame + one per border (env 0..5) */

#if defined(__HIPCC__)
#define XP_UNROLL _Pragma("unroll")
#define XP_NOUNROLL _Pragma("nounroll")
#else
#define XP_UNROLL
#define XP_NOUNROLL
#endif

struct XpFrameWork {
  union {                    /* scratch areas that are never live together */
    int32_t hyb_u[3][2][44]; /* P1: hybrid filter input of QMF bands 0..2: 12 slots of history + this frame's 32 */
    int32_t gsum[8][56];     /* P3: addends of the group sums, bands 9..63 of eight slots */
    int32_t peak[32][20];    /* P4: transient peak difference */
    uint32_t dl[32][13];     /* P5/P7: rounded samples of QMF bands 23..34 (the 14-slot delay looks 14 slots back); column
                                12 takes the other lanes' stores, so that the store needs no predicate */
  };
  int32_t hyb_l[32][20];     /* left hybrid sub-band samples of every slot: re 0..9 | im 10..19 */
  union {
    int32_t binpw[32][20];   /* P3: bin powers; P4: smoothed energy ... */
    int16_t ratio[32][20];   /* ... compacted in place into the transient ratios (entry i lands inside entry i / 2) */
  };
  uint32_t ap_h[32][11];     /* outputs of the hybrid sub-bands' all-pass chains (re, im pairs); column 10: the other lanes' */
  int16_t seg_h[XP_MAX_SEG][4][24]; /* per segment and group: H11, H12, H21, H22 before the segment's first slot */
  int16_t seg_d[XP_MAX_SEG][4][24]; /* per-slot increments */
};

FX_HD int xp_popc(uint32_t v) { return __builtin_popcount(v); }
FX_HD int xp_clz(uint32_t v) { return __builtin_clz(v); } /* v != 0 */
FX_HD uint32_t xp_pack16(int16_t lo, int16_t hi) { return (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16); }
FX_HD int16_t xp_lo16(uint32_t v) { return (int16_t)(v & 0xffffu); }
FX_HD int16_t xp_hi16(uint32_t v) { return (int16_t)(v >> 16); }

/* a.lo * b.lo + a.hi * b.hi of two packed pairs of int16, wrapping: one v_dot2_i32_i16 on the GPU */
FX_HD int32_t xp_dot2(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef short xp_short2 __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(xp_short2, a), __builtin_bit_cast(xp_short2, b), 0, false);
#else
  return (int32_t)((uint32_t)((int32_t)xp_lo16(a) * xp_lo16(b)) + (uint32_t)((int32_t)xp_hi16(a) * xp_hi16(b)));
#endif
}
/* a complex 16-bit rotation factor (re, im) as the two pairs the products of xp_allpass take: (re, -im) gives the real
   part, (im, re) the imaginary one.  The PS tables hold no -32768 (tests/test_tables.py), so -im is a short and the sum of
   two products of a sample and a factor stays below 2^31: the reference's saturating add / subtract never saturates
   there, and the wrapping dot product is the same number. */
struct XpPhase {
  uint32_t re_pair, im_pair;
};
FX_HD XpPhase xp_phase_pairs(int16_t re, int16_t im) {
  XpPhase p;
  p.re_pair = xp_pack16(re, (int16_t)-im);
  p.im_pair = xp_pack16(im, re);
  return p;
}
/* xp_allpass (sbr_ps.h, ps_dec.c:236 / :339) on packed (re, im) pairs: d0 = the 2-slot line's oldest entry (replaced by the
   new sample), e0 / e1 / e2 = the three links' entries at their read positions (replaced); returns the chain's output pair */
FX_HD uint32_t xp_allpass_packed(uint32_t &d0, uint32_t new_pair, const XpPhase &ph, uint32_t &e0, uint32_t &e1, uint32_t &e2,
                                 const XpPhase &p0, const XpPhase &p1, const XpPhase &p2, int16_t decay0, int16_t decay1,
                                 int16_t decay2) {
  int16_t in_re = (int16_t)(xp_dot2(d0, ph.re_pair) >> 15), in_im = (int16_t)(xp_dot2(d0, ph.im_pair) >> 15);
  d0 = new_pair;
  uint32_t *e[3] = {&e0, &e1, &e2};
  const XpPhase *pp[3] = {&p0, &p1, &p2};
  const int16_t decay[3] = {decay0, decay1, decay2};
  XP_UNROLL
  for (int m = 0; m < 3; m++) {
    const uint32_t s = *e[m];
    int16_t t_re = (int16_t)(xp_dot2(s, pp[m]->re_pair) >> 15), t_im = (int16_t)(xp_dot2(s, pp[m]->im_pair) >> 15);
    t_re = (int16_t)(t_re - xs_mult16_shl(in_re, decay[m]));
    t_im = (int16_t)(t_im - xs_mult16_shl(in_im, decay[m]));
    *e[m] = xp_pack16((int16_t)(in_re + xs_mult16_shl(t_re, decay[m])), (int16_t)(in_im + xs_mult16_shl(t_im, decay[m])));
    in_re = t_re;
    in_im = t_im;
  }
  return xp_pack16(in_re, in_im);
}

/* env_calc.c:1099 on one word, without a branch: a left count and a right count of which at most one is not zero (a shift
   by zero is the identity either way).  Inside the slot walks the counts are lane constants; as three-way branches per word
   the same thing cost a dozen scalar instructions around two vector ones. */
FX_HD int32_t xp_adj_word(int32_t v, int shift) {
  if (shift > 31) shift = 31;
  if (shift < -31) shift = -31;
  const int shl = shift > 0 ? shift : 0, shr = shift < 0 ? -shift : 0;
  return (int32_t)((uint32_t)v << shl) >> shr;
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
                      int32_t *xr, int lb_scale, int ov_lb_scale, int hb_scale, int st_syn, int lsb, int usb,
                      int ps_scale_done = XP_NO_PS_SCALE) {
#if defined(__HIP_DEVICE_COMPILE__)
  /* The matrix rows this frame reads before anything rewrites them, fetched now: P1's look-ahead words of QMF bands
     0..2 (lane = slot) and P3's 32 slots of the lane's band.  Issued ahead of the state rescale and the hybrid
     filters, their memory latency is covered instead of being paid once per phase and per group of eight slots. */
  int32_t p1v[3][2], p3re[32], p3im[32];
  {
    const int l1 = cx.lane & 31;
    XP_UNROLL
    for (int b = 0; b < 3; b++) {
      p1v[b][0] = xl[(l1 + 6) * 128 + b];
      p1v[b][1] = xl[(l1 + 6) * 128 + 64 + b];
    }
    XP_UNROLL
    for (int l = 0; l < 32; l++) {
      p3re[l] = xl[l * 128 + cx.lane];
      p3im[l] = xl[l * 128 + 64 + cx.lane];
    }
  }
#endif
  /* sbr_dec.c:1252.  The GPU kernel has been through it already, on the state words in registers on their way into LDS
     (sbr_ps_kernel.hip: xp_init_ps_scale_regs), and hands the result in */
  const int ps_scale = ps_scale_done != XP_NO_PS_SCALE ? ps_scale_done : xp_init_ps_scale(cx, ps, lb_scale, ov_lb_scale, hb_scale);
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
#if defined(__HIP_DEVICE_COMPILE__)
        int32_t v = xp_adj_word(p1v[b][c], sha); /* lane = slot l */
#else
        int32_t v = xp_adj_word(xl[(l + 6) * 128 + 64 * c + b], sha);
#endif
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
  const int usb_prev = cx.uni(ps->usb);
  uint32_t seg_mask = 0; /* bit l: a border was reached at slot l -- segment popcount(bits 0..l) starts there (a scalar) */
  int clear_slot = 32; /* first slot that runs with the new usb: the slot of border 0, if it is reached at all */
  {
    int env = 0, nseg = 1;
    int next = cx.uni(pf->border_position[0]); /* the border the counter is waiting for; 64 = none left */
    for (int l = 0; l < 32; l++) {
      if (l == next) {
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
        seg_mask |= 1u << l;
        nseg++;
        env++;
        next = env <= XAAC_PS_MAX_ENV ? cx.uni(pf->border_position[env]) : 64;
      }
    }
  }
  cx.sync();
  XP_T(3);

  /* ---- P3: band powers (ps_dec.c:520-545).  A lane walks its band through eight slots at a time: the sixteen row
     words of the next eight are in flight while these are worked on. */
  {
    XP_UNROLL
    for (int c = 0; c < 4; c++) {
      XS_PAR(sb, 0, 64) {
        int32_t rre[8], rim[8];
        XP_UNROLL
        for (int ls = 0; ls < 8; ls++) {
#if defined(__HIP_DEVICE_COMPILE__)
          rre[ls] = p3re[8 * c + ls]; /* lane = band sb */
          rim[ls] = p3im[8 * c + ls];
#else
          rre[ls] = xl[(8 * c + ls) * 128 + sb];
          rim[ls] = xl[(8 * c + ls) * 128 + 64 + sb];
#endif
        }
        const int gsh = sb < 11 ? 0 : (sb < 18 ? 1 : (sb < 23 ? 2 : (sb < 35 ? 3 : 4))); /* group_shift of the band's group */
        /* where the band's power goes: bands 3..8 are bins of their own, bands 9.. addends of the group sums, bands 0..2
           (whose bins come from the hybrid sub-bands) write the spare column 55 of gsum -- one store per slot through a
           lane pointer and a lane stride instead of a tree of predicated regions */
        const bool own_bin = sb >= 3 && sb < 9;
        int32_t *dst = sb < 3 ? &w->gsum[0][55] : (own_bin ? &w->binpw[8 * c][sb + 5] : &w->gsum[0][sb - 9]);
        const int dstride = own_bin ? 20 : 56;
        XP_UNROLL
        for (int ls = 0; ls < 8; ls++) {
          const int l = 8 * c + ls;
          const int usb_l = l >= clear_slot ? usb : usb_prev;
          const int sh = sb < lsb ? (l < 6 ? ov_lb_shift : lb_shift) : (sb < usb ? hb_shift : 0);
          const int32_t re = xp_adj_word(rre[ls], sh), im = xp_adj_word(rim[ls], sh);
          const int32_t pw = xp_power(re, im);
          dst[ls * dstride] = own_bin ? pw : (sb < usb_l ? (pw >> gsh) : 0);
        }
      }
      cx.sync();
      XS_PAR(i, 0, 48) { /* bins 14..19 = sums over the groups [9,11) [11,14) [14,18) [18,23) [23,35) [35,64) */
        const int ls = i / 6, g = i % 6;
        const int b0 = T->borders_group[16 + g], b1 = T->borders_group[17 + g];
        int32_t acc = 0;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(XP_OLD_GSUM)
        /* the same saturating adds in the same order, eight addends fetched at a time (a group has up to 29: one LDS
           latency per eight instead of one per addend; slots past the group's end re-read its last word and add zero) */
        for (int sb = b0; sb < b1; sb += 8) {
          int32_t t[8];
          XP_UNROLL
          for (int j = 0; j < 8; j++) t[j] = w->gsum[ls][(sb + j < b1 ? sb + j : b1 - 1) - 9];
          XP_UNROLL
          for (int j = 0; j < 8; j++) acc = fx_add_sat(acc, sb + j < b1 ? t[j] : 0);
        }
#else
        for (int sb = b0; sb < b1; sb++) acc = fx_add_sat(acc, w->gsum[ls][sb - 9]);
#endif
        w->binpw[8 * c + ls][14 + g] = acc;
      }
      cx.sync();
    }
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
    int32_t pin[32]; /* the bin's 32 powers first: the recursion below then waits for no LDS load */
    XP_UNROLL
    for (int l = 0; l < 32; l++) pin[l] = w->binpw[l][bin];
    XP_UNROLL
    for (int l = 0; l < 32; l++) {
      int32_t pw = fx_shl(pin[l], 1);
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

  /* ---- P5 + P7, one walk over the slots per lane.
     All-pass chains (ps_dec.c:236 hybrid sub-bands, :339 QMF bands 3..22 whatever usb is): the chain of QMF band sb
     runs on lane sb -- its input is the lane's own row word, its output feeds the lane's own rotation --, the chains
     of the ten hybrid sub-bands on lanes 32..41 (their outputs go to LDS for P6).  A chain's delay lines (a 2-slot
     line and three links of 3, 4 and 5 slots) are loaded oldest-first into registers, shifted by renaming in the
     unrolled loop, and stored back at the end: no LDS traffic inside the recursion.
     Delays, rotation, output (ps_dec.c:602-648, :893-945, generic:1610): the 14-slot delay reads the lane's own
     registers of slot l - 14, the 1-slot delay and the interpolated coefficients of the band's group stay in
     registers too.  Bands 0..2 are written by P6. */
  {
    const int idx0 = cx.uni(ps->idx), idx_long0 = cx.uni(ps->idx_long);
    const int is0 = cx.uni(ps->idx_ser[0]), is1 = cx.uni(ps->idx_ser[1]), is2 = cx.uni(ps->idx_ser[2]);
    const int clear_lo = (usb > usb_prev && usb_prev) ? usb_prev : 64; /* ps_dec.c:733-757: bands that just became active */
    const int clear_hi = usb < 23 ? usb : 23;
    XS_PAR(sb, 0, 64) {
      /* -- the lane's chain, if it has one */
      const int qmf_chain = sb >= 3 && sb < 23, hyb_chain = sb >= 32 && sb < 42, chain = qmf_chain || hyb_chain;
      const int csb = qmf_chain ? sb : (hyb_chain ? sb - 32 : 3);
      const int di = 9 + 3 * (csb - 3);
      const int16_t *ph = hyb_chain ? &T->frac_delay_phase_fac_qmf_sub_re_im[2 * csb] : &T->frac_delay_phase_fac_qmf_re_im[2 * csb];
      const int16_t *pser = hyb_chain ? &T->frac_delay_phase_fac_qmf_sub_ser_re_im[2 * csb] : &T->frac_delay_phase_fac_qmf_ser_re_im[2 * csb];
      const int pstep = hyb_chain ? 32 : 64;
      const XpPhase phase = xp_phase_pairs(ph[0], ph[1]);
      const XpPhase ps0 = xp_phase_pairs(pser[0], pser[1]), ps1 = xp_phase_pairs(pser[pstep], pser[pstep + 1]),
                    ps2 = xp_phase_pairs(pser[2 * pstep], pser[2 * pstep + 1]);
      const int16_t dec0 = hyb_chain ? T->rev_link_decay_ser[0] : T->decay_scale_factor[qmf_chain ? di : 9];
      const int16_t dec1 = hyb_chain ? T->rev_link_decay_ser[1] : T->decay_scale_factor[qmf_chain ? di + 1 : 10];
      const int16_t dec2 = hyb_chain ? T->rev_link_decay_ser[2] : T->decay_scale_factor[qmf_chain ? di + 2 : 11];
      uint32_t d0[2] = {0, 0}, r0[3] = {0, 0, 0}, r1[4] = {0, 0, 0, 0}, r2[5] = {0, 0, 0, 0, 0}; /* oldest first */
      if (chain) {
        XP_UNROLL
        for (int j = 0; j < 2; j++) {
          const int16_t *q = hyb_chain ? &ps->sub[(idx0 + j) % 2][2 * csb] : &ps->ap[(idx0 + j) % 2][2 * csb];
          d0[j] = xp_pack16(q[0], q[1]);
        }
        XP_UNROLL
        for (int j = 0; j < 3; j++) {
          const int16_t *q = hyb_chain ? &ps->sub_ser[(is0 + j) % 3][0][2 * csb] : &ps->ser[(is0 + j) % 3][0][2 * csb];
          r0[j] = xp_pack16(q[0], q[1]);
        }
        XP_UNROLL
        for (int j = 0; j < 4; j++) {
          const int16_t *q = hyb_chain ? &ps->sub_ser[(is1 + j) % 4][1][2 * csb] : &ps->ser[(is1 + j) % 4][1][2 * csb];
          r1[j] = xp_pack16(q[0], q[1]);
        }
        XP_UNROLL
        for (int j = 0; j < 5; j++) {
          const int16_t *q = hyb_chain ? &ps->sub_ser[(is2 + j) % 5][2][2 * csb] : &ps->ser[(is2 + j) % 5][2][2 * csb];
          r2[j] = xp_pack16(q[0], q[1]);
        }
      }
      /* -- the lane's band */
      const int g = T->band_to_group[sb];
      const int bin_sb = sb < 23 ? T->delay_to_bin[sb] : (sb < 35 ? 18 : 19); /* the band's transient-detector bin */
      /* segment 0 continues the last frame's interpolation */
      int16_t h11 = w->seg_h[0][0][g], h12 = w->seg_h[0][1][g], h21 = w->seg_h[0][2][g], h22 = w->seg_h[0][3][g];
      int16_t d11 = w->seg_d[0][0][g], d12 = w->seg_d[0][1][g], d21 = w->seg_d[0][2][g], d22 = w->seg_d[0][3][g];
      uint32_t prev = sb >= 35 ? xp_pack16(ps->sd[2 * (sb - 35)], ps->sd[2 * (sb - 35) + 1]) : 0u;
      /* The walk: four slots per pass of the loop (the delay lines of 2 and 4 slots are back in place after four
         steps, the 3- and 5-slot ones cost a few moves), the next pass's eight row words and the next slot's LDS
         operands in flight meanwhile.  The body has no lane-dependent branches: every lane runs the chain arithmetic
         (on don't-care values where it has no chain), band classes are selects, only the stores are predicated. */
      const int is_ap = sb < 23, is_d14 = sb >= 23 && sb < 35;
      /* the band's scale shift as the two counts of xp_adj_word, for the overlap slots (l < 6) and for the others: lane
         constants, chosen per slot by a uniform condition (worked out per word they were a dozen instructions of every slot) */
      int shl_ov, shr_ov, shl_lb, shr_lb;
      {
        const int s_ov = sb < lsb ? ov_lb_shift : (sb < usb ? hb_shift : 0), s_lb = sb < lsb ? lb_shift : (sb < usb ? hb_shift : 0);
        const int c_ov = s_ov > 31 ? 31 : (s_ov < -31 ? -31 : s_ov), c_lb = s_lb > 31 ? 31 : (s_lb < -31 ? -31 : s_lb);
        shl_ov = c_ov > 0 ? c_ov : 0, shr_ov = c_ov < 0 ? -c_ov : 0;
        shl_lb = c_lb > 0 ? c_lb : 0, shr_lb = c_lb < 0 ? -c_lb : 0;
      }
      const int ldj = is_d14 ? 2 * (sb - 23) : 0, dlj = is_d14 ? sb - 23 : 12, apj = hyb_chain ? csb : 10;
      int16_t tr_nx = w->ratio[0][bin_sb];
      int32_t hre_nx = w->hyb_l[0][csb], him_nx = w->hyb_l[0][10 + csb];
      uint32_t ld_nx = xp_pack16(ps->ld[idx_long0 % 14][ldj], ps->ld[idx_long0 % 14][ldj + 1]);
      int32_t nre[4], nim[4];
      XP_UNROLL
      for (int j = 0; j < 4; j++) {
        nre[j] = xl[j * 128 + sb];
        nim[j] = xl[j * 128 + 64 + sb];
      }
      XP_NOUNROLL
      for (int l0 = 0; l0 < 32; l0 += 4) {
        int32_t cre[4], cim[4];
        XP_UNROLL
        for (int j = 0; j < 4; j++) {
          cre[j] = nre[j];
          cim[j] = nim[j];
        }
        if (l0 + 4 < 32) {
          XP_UNROLL
          for (int j = 0; j < 4; j++) {
            nre[j] = xl[(l0 + 4 + j) * 128 + sb];
            nim[j] = xl[(l0 + 4 + j) * 128 + 64 + sb];
          }
        }
        XP_UNROLL
        for (int j = 0; j < 4; j++) {
          const int l = l0 + j;
          const int usb_l = l >= clear_slot ? usb : usb_prev;
          const int16_t tr = tr_nx;
          const int32_t hre = hre_nx, him = him_nx;
          const uint32_t ld_cur = ld_nx;
          {
            const int ln = l + 1 < 32 ? l + 1 : 31, pn = (idx_long0 + ln) % 14;
            tr_nx = w->ratio[ln][bin_sb];
            hre_nx = w->hyb_l[ln][csb];
            him_nx = w->hyb_l[ln][10 + csb];
            ld_nx = xp_pack16(ps->ld[pn][ldj], ps->ld[pn][ldj + 1]);
          }
          const int shl = l < 6 ? shl_ov : shl_lb, shr = l < 6 ? shr_ov : shr_lb;
          const int32_t re0 = (int32_t)((uint32_t)cre[j] << shl) >> shr, im0 = (int32_t)((uint32_t)cim[j] << shl) >> shr;
          const int16_t q_re = fx_round16(re0), q_im = fx_round16(im0);
          const uint32_t q = xp_pack16(q_re, q_im);
          /* the chain */
          if (l == clear_slot) { /* (uniform) the three links' lines of the bands that just became active */
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile(""); /* keeps this a scalar branch taken once per frame: as twelve selects it ran in every slot */
#endif
            const int c = qmf_chain && sb >= clear_lo && sb < clear_hi;
            XP_UNROLL
            for (int m = 0; m < 3; m++) r0[m] = c ? 0u : r0[m];
            XP_UNROLL
            for (int m = 0; m < 4; m++) r1[m] = c ? 0u : r1[m];
            XP_UNROLL
            for (int m = 0; m < 5; m++) r2[m] = c ? 0u : r2[m];
          }
          const uint32_t in_pair = hyb_chain ? xp_pack16(fx_round16(hre), fx_round16(him)) : q;
          uint32_t dv = d0[0], e0 = r0[0], e1 = r1[0], e2 = r2[0];
          const uint32_t o_chain = xp_allpass_packed(dv, in_pair, phase, e0, e1, e2, ps0, ps1, ps2, dec0, dec1, dec2);
          d0[0] = d0[1];
          d0[1] = dv;
          r0[0] = r0[1]; r0[1] = r0[2];
          r0[2] = e0;
          r1[0] = r1[1]; r1[1] = r1[2]; r1[2] = r1[3];
          r1[3] = e1;
          r2[0] = r2[1]; r2[1] = r2[2]; r2[2] = r2[3]; r2[3] = r2[4];
          r2[4] = e2;
          w->ap_h[l][apj] = o_chain; /* lanes without a hybrid chain write the spare column */
          /* the interpolated coefficients of the band's group */
          if ((seg_mask >> l) & 1u) { /* (uniform) a border: they restart from the old targets */
            const int s = xp_popc(seg_mask & (0xffffffffu >> (31 - l)));
            h11 = w->seg_h[s][0][g]; h12 = w->seg_h[s][1][g]; h21 = w->seg_h[s][2][g]; h22 = w->seg_h[s][3][g];
            d11 = w->seg_d[s][0][g]; d12 = w->seg_d[s][1][g]; d21 = w->seg_d[s][2][g]; d22 = w->seg_d[s][3][g];
          }
          h11 = (int16_t)(h11 + d11); /* the interpolation advances whether or not the band is rotated */
          h12 = (int16_t)(h12 + d12);
          h21 = (int16_t)(h21 + d21);
          h22 = (int16_t)(h22 + d22);
          /* the decorrelated sample of the band: all-pass output, or the input of 14 slots / 1 slot ago.  The 14-slot
             line holds what slot l - 14 put in if that slot ran with the band active, else what the state held. */
          const int active = sb < usb_l;
          const int fed14 = l >= 14 && sb < (l - 14 >= clear_slot ? usb : usb_prev);
          /* (both candidates are read by every lane -- a lane without a delay line reads entries it does not use -- and
             the value is a select: no predicated region inside the walk) */
          const uint32_t o14_new = w->dl[l >= 14 ? l - 14 : 0][dlj];
          const uint32_t o14_old = l < 14 ? ld_cur : xp_pack16(ps->ld[(idx_long0 + l) % 14][ldj], ps->ld[(idx_long0 + l) % 14][ldj + 1]);
          const uint32_t o14 = fed14 ? o14_new : o14_old;
          const uint32_t o = is_ap ? o_chain : (is_d14 ? o14 : prev);
          w->dl[l][dlj] = q; /* lanes without a 14-slot line write the spare column */
          prev = active ? q : prev;
          int32_t re = re0, im = im0;
          int32_t r_re = xp_m16x16_shl(xp_lo16(o), tr), r_im = xp_m16x16_shl(xp_hi16(o), tr);
          xp_rotate(&re, &r_re, h11, h12, h21, h22);
          xp_rotate(&im, &r_im, h11, h12, h21, h22);
          re = active ? re : re0; /* above usb: the left sample passes, the right one is zero */
          im = active ? im : im0;
          r_re = active ? r_re : 0;
          r_im = active ? r_im : 0;
          if (common_shift < 0) {
            const int cs = -common_shift > 31 ? 31 : -common_shift;
            re = fx_shr(re, cs);
            im = fx_shr(im, cs);
          } else if (common_shift > 0) {
            re = fx_shl_sat(re, common_shift);
            im = fx_shl_sat(im, common_shift);
          }
          if (sb >= 3) {
            xl[l * 128 + sb] = re;
            xl[l * 128 + 64 + sb] = im;
            xr[l * 128 + sb] = r_re;
            xr[l * 128 + 64 + sb] = r_im;
          }
        }
      }
      /* -- the delay lines as the slot loop leaves them */
      if (chain) {
        XP_UNROLL
        for (int j = 0; j < 2; j++) {
          int16_t *q = hyb_chain ? &ps->sub[(idx0 + j) % 2][2 * csb] : &ps->ap[(idx0 + j) % 2][2 * csb]; /* 32 slots: same phase */
          q[0] = xp_lo16(d0[j]);
          q[1] = xp_hi16(d0[j]);
        }
        XP_UNROLL
        for (int j = 0; j < 3; j++) {
          int16_t *q = hyb_chain ? &ps->sub_ser[(is0 + 32 + j) % 3][0][2 * csb] : &ps->ser[(is0 + 32 + j) % 3][0][2 * csb];
          q[0] = xp_lo16(r0[j]);
          q[1] = xp_hi16(r0[j]);
        }
        XP_UNROLL
        for (int j = 0; j < 4; j++) {
          int16_t *q = hyb_chain ? &ps->sub_ser[(is1 + 32 + j) % 4][1][2 * csb] : &ps->ser[(is1 + 32 + j) % 4][1][2 * csb];
          q[0] = xp_lo16(r1[j]);
          q[1] = xp_hi16(r1[j]);
        }
        XP_UNROLL
        for (int j = 0; j < 5; j++) {
          int16_t *q = hyb_chain ? &ps->sub_ser[(is2 + 32 + j) % 5][2][2 * csb] : &ps->ser[(is2 + 32 + j) % 5][2][2 * csb];
          q[0] = xp_lo16(r2[j]);
          q[1] = xp_hi16(r2[j]);
        }
      }
      if (sb >= 35) { /* the 1-slot delay line as the last slot leaves it */
        ps->sd[2 * (sb - 35)] = xp_lo16(prev);
        ps->sd[2 * (sb - 35) + 1] = xp_hi16(prev);
      }
      if (is_d14) { /* each position of the 14-slot ring ends up with the input of the last slot that wrote it */
        for (int l = (clear_slot == 0 || usb == usb_prev) ? 18 : 0; l < 32; l++) {
          const int pos = (idx_long0 + l) % 14;
          if (sb < (l >= clear_slot ? usb : usb_prev)) {
            ps->ld[pos][ldj] = xp_lo16(w->dl[l][dlj]);
            ps->ld[pos][ldj + 1] = xp_hi16(w->dl[l][dlj]);
          }
        }
      }
    }
    cx.sync();
    XS_ONE {
      ps->idx = (int16_t)idx0; /* 32 slots later the 2-slot line is in the same phase */
      ps->idx_ser[0] = (int16_t)((is0 + 32) % 3);
      ps->idx_ser[1] = (int16_t)((is1 + 32) % 4);
      ps->idx_ser[2] = (int16_t)((is2 + 32) % 5);
      ps->idx_long = (int16_t)((idx_long0 + 32) % 14);
      if (clear_slot < 32) ps->usb = (int16_t)usb;
    }
  }
  cx.sync();
  XP_T(6);

  /* ---- P6: rotation of the hybrid sub-bands (ps_dec.c:856, groups 0..9) and hybrid synthesis of QMF bands 0..2
     (the saturating sums of ps_dec.c:899-925, in sub-band order), one (slot, band, re | im) per lane */
  XS_PAR(i, 0, 192) {
    const int l = i / 6, b = (i % 6) >> 1, c = i & 1;
    const int p = b == 0 ? 0 : 4 + 2 * b, n = b == 0 ? 6 : 2;
    const uint32_t below = seg_mask & (0xffffffffu >> (31 - l));     /* borders at or before slot l */
    const int s = xp_popc(below), nn = l - (below ? 31 - xp_clz(below) : 0) + 1; /* slots since the segment began */
    int32_t acc_l = 0, acc_r = 0;
    for (int k = 0; k < n; k++) {
      const int sb = p + k;
      const int16_t h11 = (int16_t)(w->seg_h[s][0][sb] + nn * w->seg_d[s][0][sb]);
      const int16_t h12 = (int16_t)(w->seg_h[s][1][sb] + nn * w->seg_d[s][1][sb]);
      const int16_t h21 = (int16_t)(w->seg_h[s][2][sb] + nn * w->seg_d[s][2][sb]);
      const int16_t h22 = (int16_t)(w->seg_h[s][3][sb] + nn * w->seg_d[s][3][sb]);
      const int16_t tr = w->ratio[l][T->hybrid_to_bin[sb]];
      const uint32_t o = w->ap_h[l][sb];
      int32_t lv = w->hyb_l[l][10 * c + sb];
      int32_t rv = xp_m16x16_shl(c ? xp_hi16(o) : xp_lo16(o), tr);
      xp_rotate(&lv, &rv, h11, h12, h21, h22);
      acc_l = k == 0 ? lv : fx_add_sat(acc_l, lv);
      acc_r = k == 0 ? rv : fx_add_sat(acc_r, rv);
    }
    if (common_shift < 0)
      acc_l = fx_shr(acc_l, -common_shift > 31 ? 31 : -common_shift);
    else if (common_shift > 0)
      acc_l = fx_shl_sat(acc_l, common_shift);
    xl[l * 128 + 64 * c + b] = acc_l;
    xr[l * 128 + 64 * c + b] = acc_r;
  }
  {
    const int s = xp_popc(seg_mask), n = 32 - (seg_mask ? 31 - xp_clz(seg_mask) : 0);
    XS_PAR(g, 0, XAAC_PS_GROUPS) {
      ps->H11_H12[2 * g] = (int16_t)(w->seg_h[s][0][g] + n * w->seg_d[s][0][g]);
      ps->H11_H12[2 * g + 1] = (int16_t)(w->seg_h[s][1][g] + n * w->seg_d[s][1][g]);
      ps->H21_H22[2 * g] = (int16_t)(w->seg_h[s][2][g] + n * w->seg_d[s][2][g]);
      ps->H21_H22[2 * g + 1] = (int16_t)(w->seg_h[s][3][g] + n * w->seg_d[s][3][g]);
    }
  }
  cx.sync();
  XP_T(7);
  return ps_scale;
}

#endif /* XAAC_SBR_PS_FRAME_H */
