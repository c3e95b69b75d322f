/*
 * esbr_ps.h -- the float parametric-stereo tool of the reference's default SBR path (ixheaacd_esbr_apply_ps,
 * decoder/ixheaacd_ps_dec_flt.c:389: hybrid analysis :119 / :75, decorrelation :510, rotation :841, hybrid synthesis
 * :203), shared by the gfx950 kernel (esbr_ps_kernel.hip: one wave = one stream-frame) and, compiled for the host with
 * one "lane", by the checker (oracle/oracle_esbr.cpp).
 *
 * What the reference's AAC parser can hand this tool is narrower than the tool: ixheaacd_read_ps_data forces the
 * 20-band configuration and the plain (non-PCA) mixing rule (sbrdec_lpfuncs.c:636-637), nothing ever writes the IPD / OPD
 * index maps, and the tool is created with ps_mode 0.  This restatement covers exactly that: 20 bands; the mixing
 * matrix from the table tools/gen_tables_esbr_ps.py makes with the C library's cos / sin; the phase step with all
 * indices zero, which leaves h_re * 1.0f and h_im = h_re * 0.0f (a signed zero) below the IPD band limit.
 *
 * Float results depend on the order of every operation: expressions keep the reference's operand order, sums run in its
 * order (the powers of a bin accumulate group by group, sub-band by sub-band), the mixing matrix is stepped slot by slot.
 */
#ifndef XAAC_ESBR_PS_H
#define XAAC_ESBR_PS_H

#include "esbr_core.h"

#if defined(__HIPCC__)
#define XAAC_TAB_QUAL static __device__ const
#include "tables_esbr_ps.inc"
#undef XAAC_TAB_QUAL
#else
#include "tables_esbr_ps.inc"
#endif

#pragma clang fp contract(off)

struct XfWork {
  float hl_re[32][12], hl_im[32][12], hr_re[32][12], hr_im[32][12]; /* hyb_left_* / hyb_right_*: 12 hybrid sub-bands */
  union {
    float pw[32][20];                                               /* pow_arr */
    float hin[2][3][44];                                            /* before the band powers: the hybrid filters' work buffer
                                                                       (re | im) of QMF bands 0..2: 12 slots of history, then
                                                                       rows 6..37 of the left matrix */
    float hv[XAAC_PS_MAX_ENV][8][20];                               /* behind the transient detector: h11_re_vec ... h22_im_vec
                                                                       of every envelope */
  };
  float tr[32][20];                                                 /* trans_ratio_arr */
};

#define XF_NEG 0x1000 /* NEGATE_IPD_MASK */
#ifndef XF_WALK_CH
#define XF_WALK_CH 4 /* rows in flight in the decorrelation + rotation walk (four arrays of them beside the chains' state) */
#endif

/* one all-pass chain step (ps_dec_flt.c:693-716): in = the delayed, phase-rotated sample; returns the chain's output */
FX_HD void xf_allpass(float &r_r0, float &i_r0, float *ser_re, float *ser_im /* [3] current ring cells */,
                      const float *pf_re, const float *pf_im /* [3] */, float decay, const float *link /* [3] */) {
  for (int m = 0; m < 3; m++) {
    const float real0 = ser_re[m], imag0 = ser_im[m];
    float real = real0 * pf_re[m] - imag0 * pf_im[m];
    float imag = real0 * pf_im[m] + imag0 * pf_re[m];
    real += -decay * link[m] * r_r0;
    imag += -decay * link[m] * i_r0;
    ser_re[m] = r_r0 + decay * link[m] * real;
    ser_im[m] = i_r0 + decay * link[m] * imag;
    r_r0 = real;
    i_r0 = imag;
  }
}

/* L: the left matrix, rows 0..37 (rows 32..37: bands 0..4 of the next frame's first slots, sbr_dec.c:487-505); R: the
   right matrix, rows 0..31.  usb = sub_band_end. */
FX_HD void xf_apply_ps(const XsCx &cx, const xaac_ps_frame *pf, xaac_esbr_ps_state *ps, XfWork *w, const XeMat &L,
                       const XeMat &R, int usb) {
  const int num_env = pf->num_env;
  const int k0 = pf->border_position[0], k1 = pf->border_position[num_env];
  const int32_t *gb = xaac_eps_group_borders_20_tbl, *gmap = xaac_eps_bin_group_map_20;
  /* bands above the SBR range start from silence (ps_dec_flt.c:427-438) */
  XS_PAR(sb, usb < 0 ? 0 : usb, 64) {
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < 5; k++) { /* delay_sample_ser = 3, 4, 5 */
        if (k < 3 + i) {
          ps->ser_qmf_re[i][k][sb] = 0;
          ps->ser_qmf_im[i][k][sb] = 0;
        }
      }
    for (int k = 0; k < 14; k++) {
      ps->qmf_delay_re[k][sb] = 0;
      ps->qmf_delay_im[k][sb] = 0;
    }
  }
  XE_T(0);
  /* hybrid analysis of QMF bands 0..2 (8 + 2 + 2 sub-bands); the 12-slot history is the filter's past.  The three columns
     are staged once (a lane per (band, row): as 78 column gathers per slot-lane they were 78 uncoalesced loads each), then a
     lane takes a slot and half of the sub-bands: band 0's channels 0..3 and band 1, or channels 4..7 and band 2. */
  XS_PAR(e, 0, 96) {
    const int band = e >> 5, r = e & 31;
    w->hin[0][band][12 + r] = L.r(6 + r, band);
    w->hin[1][band][12 + r] = L.i(6 + r, band);
  }
  XS_PAR(j, 0, 36) {
    const int band = j / 12, n = j % 12;
    w->hin[0][band][n] = ps->hyb_hist_re[band][n];
    w->hin[1][band][n] = ps->hyb_hist_im[band][n];
  }
  cx.sync();
  XS_PAR(e, 0, 64) {
    const int i = e & 31, half = e >> 5;
    {
      const float *wr = &w->hin[0][0][i], *wi = &w->hin[1][0][i];
      for (int q = 4 * half; q < 4 * half + 4; q++) {
        float real = 0, imag = 0;
        for (int n = 0; n < 13; n++) {
          const float c = xaac_eps_cos_sin_mod_8channel[26 * q + 2 * n], sn = xaac_eps_cos_sin_mod_8channel[26 * q + 2 * n + 1];
          real += xaac_eps_p8_13_20[n] * (wr[n] * c - wi[n] * sn);
          imag += xaac_eps_p8_13_20[n] * (wi[n] * c + wr[n] * sn);
        }
        w->hl_re[i][q] = real;
        w->hl_im[i][q] = imag;
      }
    }
    {
      const int band = 1 + half, ch_offset = 8 + 2 * half;
      const float *wr = &w->hin[0][band][i], *wi = &w->hin[1][band][i];
      for (int q = 0; q < 2; q++) {
        float real = 0, imag = 0;
        for (int n = 0; n < 13; n++) {
          const float c = xaac_eps_cos_mod_2channel[13 * q + n];
          real += xaac_eps_p2_13_20[n] * (wr[n] * c);
          imag += xaac_eps_p2_13_20[n] * (wi[n] * c);
        }
        w->hl_re[i][ch_offset + q] = real;
        w->hl_im[i][ch_offset + q] = imag;
      }
    }
  }
  cx.sync();
  XS_PAR(i, 0, 32) {
    /* ps_dec_flt.c:446-459 */
    w->hl_re[i][3] += w->hl_re[i][4];
    w->hl_im[i][3] += w->hl_im[i][4];
    w->hl_re[i][4] = 0.;
    w->hl_im[i][4] = 0.;
    w->hl_re[i][2] += w->hl_re[i][5];
    w->hl_im[i][2] += w->hl_im[i][5];
    w->hl_re[i][5] = 0.;
    w->hl_im[i][5] = 0.;
    for (int q = 0; q < 12; q++) {
      w->hr_re[i][q] = 0;
      w->hr_im[i][q] = 0;
    }
  }
  XS_PAR(j, 0, 36) { /* the next frame's history: the last 12 work slots = rows 26..37 of L */
    const int band = j / 12, n = j % 12;
    ps->hyb_hist_re[band][n] = w->hin[0][band][32 + n];
    ps->hyb_hist_im[band][n] = w->hin[1][band][32 + n];
  }
  cx.sync(); /* (the band powers below take the work buffer's place) */
  XE_T(1);
  /* band powers per parameter bin (:606-632), lane = slot; a bin's sum runs group by group, sub-band by sub-band */
  XS_PAR(k, 0, 32) {
    for (int bin = 0; bin < 20; bin++) w->pw[k][bin] = 0;
    if (k >= k0 && k < k1) {
      for (int gr = 0; gr < 22; gr++) {
        const int bin = gmap[gr] & ~XF_NEG;
        if (gr < 10) {
          const int sb = gb[gr];
          const float a = w->hl_re[k][sb], b = w->hl_im[k][sb];
          w->pw[k][bin] += a * a + b * b;
        } else {
          for (int sb = gb[gr]; sb < gb[gr + 1]; sb++) {
            const float a = L.r(k, sb), b = L.i(k, sb);
            w->pw[k][bin] += a * a + b * b;
          }
        }
      }
    }
  }
  cx.sync();
  XE_T(2);
  /* transient detector (:634-659), lane = bin, a recursion over the slots */
  XS_PAR(bin, 0, 20) {
    float peak = ps->peak_decay_fast[bin], pdiff = ps->prev_peak_diff[bin], nrg = ps->prev_nrg[bin];
    for (int k = k0; k < k1; k++) {
      const float q = 1.5f, p = w->pw[k][bin];
      peak *= 0.765928338364649f;
      if (peak < p) peak = p;
      float d = pdiff;
      d += (1.0f - 0.75f) * (peak - p - pdiff);
      pdiff = d;
      float e = nrg;
      e += (1.0f - 0.75f) * (p - nrg);
      nrg = e;
      w->tr[k][bin] = q * d <= e ? 1.0f : e / (q * d);
    }
    ps->peak_decay_fast[bin] = peak;
    ps->prev_peak_diff[bin] = pdiff;
    ps->prev_nrg[bin] = nrg;
  }
  cx.sync();
  XE_T(3);
  /* decorrelation: all-pass chains on the hybrid sub-bands and QMF bands 3..22, plain delays above (:667-827); every
     sub-band is its own recursion over the slots.  Lanes 0..9: hybrid groups; then lanes = QMF bands 3..63. */
  const int l_delay0 = ps->delay_buf_idx;
  const int ser0[3] = {ps->delay_buf_idx_ser[0], ps->delay_buf_idx_ser[1], ps->delay_buf_idx_ser[2]};
  int l_delay_end = l_delay0, ser_end[3] = {ser0[0], ser0[1], ser0[2]};
  /* The rings are walked as shift registers: a lane fetches its delay line and its three all-pass rings in time order
     (oldest first) before the slot loop, shifts them in registers, and writes them back in time order behind the position
     the loop ends at -- the same cells, the same values, one memory round trip instead of one per slot. */
  XS_PAR(gr, 0, 10) {
    const int sb = gb[gr], bin = gmap[gr] & ~XF_NEG;
    const float pr = xaac_eps_frac_delay_phase_fac_qmf_sub_re_20[sb], pi = xaac_eps_frac_delay_phase_fac_qmf_sub_im_20[sb];
    float pfr[3], pfi[3], link[3]; /* loop constants in registers */
    for (int m = 0; m < 3; m++) {
      pfr[m] = xaac_eps_frac_delay_phase_fac_ser_qmf_sub_re_20[3 * sb + m];
      pfi[m] = xaac_eps_frac_delay_phase_fac_ser_qmf_sub_im_20[3 * sb + m];
      link[m] = xaac_eps_all_pass_link_decay_ser[m];
    }
    float dre[2], dim[2], sre[3][5] = {{0}}, sim[3][5] = {{0}};
    for (int j = 0; j < 2; j++) {
      const int idx = (l_delay0 + j) & 1;
      dre[j] = ps->sub_delay_re[idx][sb];
      dim[j] = ps->sub_delay_im[idx][sb];
    }
    for (int m = 0; m < 3; m++)
      for (int j = 0; j < 3 + m; j++) {
        const int idx = (ser0[m] + j) % (3 + m);
        sre[m][j] = ps->ser_sub_re[m][idx][sb];
        sim[m][j] = ps->ser_sub_im[m][idx][sb];
      }
    for (int k = k0; k < k1; k++) {
      const float real0 = dre[0], imag0 = dim[0];
      dre[0] = dre[1]; dim[0] = dim[1];
      dre[1] = w->hl_re[k][sb]; dim[1] = w->hl_im[k][sb];
      float r_r0 = real0 * pr - imag0 * pi, i_r0 = real0 * pi + imag0 * pr;
      float sr[3] = {sre[0][0], sre[1][0], sre[2][0]}, si[3] = {sim[0][0], sim[1][0], sim[2][0]};
      xf_allpass(r_r0, i_r0, sr, si, pfr, pfi, 1.0f, link);
      for (int m = 0; m < 3; m++) {
        for (int j = 0; j < 2 + m; j++) {
          sre[m][j] = sre[m][j + 1];
          sim[m][j] = sim[m][j + 1];
        }
        sre[m][2 + m] = sr[m];
        sim[m][2 + m] = si[m];
      }
      const float t = w->tr[k][bin];
      w->hr_re[k][sb] = t * r_r0;
      w->hr_im[k][sb] = t * i_r0;
    }
    const int cnt = k1 > k0 ? k1 - k0 : 0;
    for (int j = 0; j < 2; j++) {
      const int idx = (l_delay0 + cnt + j) & 1;
      ps->sub_delay_re[idx][sb] = dre[j];
      ps->sub_delay_im[idx][sb] = dim[j];
    }
    for (int m = 0; m < 3; m++)
      for (int j = 0; j < 3 + m; j++) {
        const int idx = (ser0[m] + cnt + j) % (3 + m);
        ps->ser_sub_re[m][idx][sb] = sre[m][j];
        ps->ser_sub_im[m][idx][sb] = sim[m][j];
      }
  }
  XE_T(4);
  /* the envelopes' target matrices (:841-1004), all of them before the walks below: bin -> h11, h12, h21, h22 (re | im) */
  const int ipd_bins = xaac_eps_ipd_bins_tbl[pf->freq_res_ipd < 0 ? 0 : (pf->freq_res_ipd > 2 ? 2 : pf->freq_res_ipd)];
  const int steps = pf->iid_quant ? 15 : 7;
  if (num_env > XAAC_PS_MAX_ENV) return; /* (the kernel and the oracle's entry check the frame before they call) */
  XS_PAR(e, 0, 20 * num_env) {
    const int env = e / 20, bin = e % 20;
    int iid = pf->iid_par_table[env][bin], icc = pf->icc_par_table[env][bin];
    iid = iid < -steps ? -steps : (iid > steps ? steps : iid);
    icc = icc < 0 ? 0 : (icc > 7 ? 7 : icc);
    const float *m = &xaac_eps_mix[(((pf->iid_quant ? 1 : 0) * 61 + iid + 30) * 8 + icc) * 4];
    float hr[4] = {m[0], m[1], m[2], m[3]}, hi[4];
    if (bin >= ipd_bins) {
      hi[0] = hi[1] = hi[2] = hi[3] = 0.0f;
    } else { /* the phase step with every IPD / OPD index zero: cos 1, sin 0 (:970-1004) */
      for (int j = 0; j < 4; j++) {
        hi[j] = hr[j] * 0.0f;
        hr[j] *= 1.0f;
      }
    }
    for (int j = 0; j < 4; j++) {
      w->hv[env][j][bin] = hr[j];
      w->hv[env][4 + j][bin] = hi[j];
    }
  }
  cx.sync();
  /* QMF bands 3..63: decorrelation (all-pass chain or plain delay, :667-827) and the rotation by the interpolated matrix
     (:1006-1224) in one walk over the slots -- the decorrelated sample goes from the chain into the rotation in a register.  (As
     two passes the right channel's rows were written, read back and written again, and the left channel's read twice: 48 KB
     per stream-frame through memory for nothing.)  An envelope's matrix starts from the previous envelope's target (the
     state's, for the first) and steps by a constant per slot. */
  XS_PAR(sb, 3, 64) {
    int gr = 10;
    while (gr < 21 && sb >= gb[gr + 1]) gr++;
    const int bin = gmap[gr] & ~XF_NEG;
    const bool neg = (gmap[gr] & XF_NEG) != 0;
    float decay = sb <= 3 ? 1.0f : 1.0f + 3.0f * 0.05f - 0.05f * (float)sb;
    decay = decay > 0.0f ? decay : 0.0f;
    const bool plain = sb >= 23;
    const int dl = plain ? xaac_eps_qmf_delay_idx_tbl[sb] : 2; /* 14, 1 or 2 cells of qmf_delay_buf belong to this band */
    const int pos = plain ? ps->delay_qmf_idx[sb] : l_delay0;
    const float pr = xaac_eps_qmf_fract_delay_phase_factor_re[sb], pi = xaac_eps_qmf_fract_delay_phase_factor_im[sb];
    float pfr[3], pfi[3], link[3]; /* loop constants in registers */
    for (int m = 0; m < 3; m++) {
      pfr[m] = xaac_eps_qmf_ser_fract_delay_phase_factor_re[3 * sb + m];
      pfi[m] = xaac_eps_qmf_ser_fract_delay_phase_factor_im[3 * sb + m];
      link[m] = xaac_eps_all_pass_link_decay_ser[m];
    }
    float dre[14] = {0}, dim[14] = {0}, sre[3][5] = {{0}}, sim[3][5] = {{0}};
    XE_UNROLL
    for (int j = 0; j < 14; j++)
      if (j < dl) {
        int idx = pos + j;
        if (idx >= dl) idx -= dl;
        dre[j] = ps->qmf_delay_re[idx][sb];
        dim[j] = ps->qmf_delay_im[idx][sb];
      }
    if (!plain)
      for (int m = 0; m < 3; m++)
        for (int j = 0; j < 3 + m; j++) {
          const int idx = (ser0[m] + j) % (3 + m);
          sre[m][j] = ps->ser_qmf_re[m][idx][sb];
          sim[m][j] = ps->ser_qmf_im[m][idx][sb];
        }
    float Hprev[8];
    for (int j = 0; j < 8; j++) Hprev[j] = ps->h_prev[j][bin];
    for (int env = 0; env < num_env; env++) {
      const int e0 = pf->border_position[env], e1 = pf->border_position[env + 1], len = e1 - e0;
      float H[8], d[8];
      for (int j = 0; j < 8; j++) {
        const float prev = Hprev[j], cur = w->hv[env][j][bin];
        const float Hp = (j >= 4 && neg) ? -prev : prev, hc = (j >= 4 && neg) ? -cur : cur;
        H[j] = Hp;
        d[j] = (hc - Hp) / (float)len;
        Hprev[j] = cur;
      }
      XE_NOUNROLL
      for (int kc = e0; kc < e1; kc += XF_WALK_CH) { /* XF_WALK_CH rows of the left channel in, as many of both channels out per burst */
        float lr[XF_WALK_CH] = {0}, li[XF_WALK_CH] = {0}, rr[XF_WALK_CH] = {0}, ri[XF_WALK_CH] = {0};
        XE_UNROLL
        for (int jj = 0; jj < XF_WALK_CH; jj++)
          if (kc + jj < e1) {
            lr[jj] = L.r(kc + jj, sb);
            li[jj] = L.i(kc + jj, sb);
          }
        XE_UNROLL
        for (int jj = 0; jj < XF_WALK_CH; jj++)
          if (kc + jj < e1) {
            const int k = kc + jj;
            const float in_re = lr[jj], in_im = li[jj];
            const float real0 = dre[0], imag0 = dim[0];
            XE_UNROLL
            for (int j = 0; j < 13; j++) {
              dre[j] = dre[j + 1];
              dim[j] = dim[j + 1];
            }
            if (dl == 14) { dre[13] = in_re; dim[13] = in_im; }
            else if (dl == 2) { dre[1] = in_re; dim[1] = in_im; }
            else { dre[0] = in_re; dim[0] = in_im; }
            float r_r0, i_r0;
            if (plain) {
              r_r0 = real0;
              i_r0 = imag0;
            } else {
              r_r0 = real0 * pr - imag0 * pi;
              i_r0 = real0 * pi + imag0 * pr;
              float sr[3] = {sre[0][0], sre[1][0], sre[2][0]}, si[3] = {sim[0][0], sim[1][0], sim[2][0]};
              xf_allpass(r_r0, i_r0, sr, si, pfr, pfi, decay, link);
              for (int m = 0; m < 3; m++) {
                for (int j = 0; j < 2 + m; j++) {
                  sre[m][j] = sre[m][j + 1];
                  sim[m][j] = sim[m][j + 1];
                }
                sre[m][2 + m] = sr[m];
                sim[m][2 + m] = si[m];
              }
            }
            const float t = w->tr[k][bin];
            const float rre = t * r_r0, rim = t * i_r0; /* the right channel's sample in front of the rotation */
            for (int j = 0; j < 8; j++) H[j] += d[j];
            /* H[0..3] = H11r H12r H21r H22r, H[4..7] = H11i H12i H21i H22i */
            const float lre = in_re, lim = in_im;
            lr[jj] = H[0] * lre - H[4] * lim + H[2] * rre - H[6] * rim;
            li[jj] = H[4] * lre + H[0] * lim + H[6] * rre + H[2] * rim;
            rr[jj] = H[1] * lre - H[5] * lim + H[3] * rre - H[7] * rim;
            ri[jj] = H[5] * lre + H[1] * lim + H[7] * rre + H[3] * rim;
          }
        XE_UNROLL
        for (int jj = 0; jj < XF_WALK_CH; jj++)
          if (kc + jj < e1) {
            L.r(kc + jj, sb) = lr[jj];
            L.i(kc + jj, sb) = li[jj];
            R.r(kc + jj, sb) = rr[jj];
            R.i(kc + jj, sb) = ri[jj];
          }
      }
    }
    const int cnt = k1 > k0 ? k1 - k0 : 0;
    const int pos_end = (pos + cnt) % dl;
    XE_UNROLL
    for (int j = 0; j < 14; j++)
      if (j < dl) {
        int idx = pos_end + j;
        if (idx >= dl) idx -= dl;
        ps->qmf_delay_re[idx][sb] = dre[j];
        ps->qmf_delay_im[idx][sb] = dim[j];
      }
    if (!plain)
      for (int m = 0; m < 3; m++)
        for (int j = 0; j < 3 + m; j++) {
          const int idx = (ser0[m] + cnt + j) % (3 + m);
          ps->ser_qmf_re[m][idx][sb] = sre[m][j];
          ps->ser_qmf_im[m][idx][sb] = sim[m][j];
        }
    if (plain) ps->delay_qmf_idx[sb] = pos_end;
  }
  XE_T(5);
  for (int k = k0; k < k1; k++) { /* where the shared ring positions end up (:829-832) */
    if (++l_delay_end >= 2) l_delay_end = 0;
    for (int m = 0; m < 3; m++)
      if (++ser_end[m] >= 3 + m) ser_end[m] = 0;
  }
  cx.sync();
  XS_ONE {
    ps->delay_buf_idx = l_delay_end;
    for (int m = 0; m < 3; m++) ps->delay_buf_idx_ser[m] = ser_end[m];
  }
  /* rotation of the ten hybrid groups, whose samples are in LDS: six lanes per group, a sixth of an envelope's slots each.
     The matrix of a slot is the sum H_prev + d + d + ... in single precision, so a lane first steps through the slots in front
     of its own (the same additions in the same order; eight adds per slot against the forty-odd operations of a rotated slot) */
  for (int env = 0; env < num_env; env++) {
    const int e0 = pf->border_position[env], e1 = pf->border_position[env + 1], len = e1 - e0;
    const int per = len > 0 ? (len + 5) / 6 : 0;
    XS_PAR(t, 0, 60) {
      const int gr = t / 6, c = t % 6, sb = gb[gr];
      const int bin = gmap[gr] & ~XF_NEG;
      const bool neg = (gmap[gr] & XF_NEG) != 0;
      const int i0 = e0 + c * per, i1 = i0 + per < e1 ? i0 + per : e1;
      if (i0 < i1) {
        float H[8], d[8];
        for (int j = 0; j < 8; j++) {
          const float prev = env ? w->hv[env - 1][j][bin] : ps->h_prev[j][bin], cur = w->hv[env][j][bin];
          const float Hp = (j >= 4 && neg) ? -prev : prev, hc = (j >= 4 && neg) ? -cur : cur;
          H[j] = Hp;
          d[j] = (hc - Hp) / (float)len;
        }
        for (int i = e0; i < i0; i++)
          for (int j = 0; j < 8; j++) H[j] += d[j];
        for (int i = i0; i < i1; i++) {
          for (int j = 0; j < 8; j++) H[j] += d[j];
          const float lre = w->hl_re[i][sb], lim = w->hl_im[i][sb], rre = w->hr_re[i][sb], rim = w->hr_im[i][sb];
          const float o_lre = H[0] * lre - H[4] * lim + H[2] * rre - H[6] * rim;
          const float o_lim = H[4] * lre + H[0] * lim + H[6] * rre + H[2] * rim;
          const float o_rre = H[1] * lre - H[5] * lim + H[3] * rre - H[7] * rim;
          const float o_rim = H[5] * lre + H[1] * lim + H[7] * rre + H[3] * rim;
          w->hl_re[i][sb] = o_lre;
          w->hl_im[i][sb] = o_lim;
          w->hr_re[i][sb] = o_rre;
          w->hr_im[i][sb] = o_rim;
        }
      }
    }
  }
  cx.sync();
  if (num_env > 0)
    XS_PAR(bin, 0, 20)
      for (int j = 0; j < 8; j++) ps->h_prev[j][bin] = w->hv[num_env - 1][j][bin];
  cx.sync();
  XE_T(6);
  /* hybrid synthesis (:203-227): QMF bands 0..2 of both channels */
  XS_PAR(n, 0, 32) {
    int ch = 0;
    for (int band = 0; band < 3; band++) {
      const int res = band == 0 ? 8 : 2;
      float lr = 0, li = 0, rr = 0, ri = 0;
      for (int k = 0; k < res; k++) {
        lr += w->hl_re[n][ch + k];
        li += w->hl_im[n][ch + k];
        rr += w->hr_re[n][ch + k];
        ri += w->hr_im[n][ch + k];
      }
      L.r(n, band) = lr; L.i(n, band) = li;
      R.r(n, band) = rr; R.i(n, band) = ri;
      ch += res;
    }
  }
  cx.sync();
  XE_T(7);
}

#endif
