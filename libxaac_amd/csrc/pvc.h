/*
 * pvc.h -- the PVC (predictive vector coding) envelope decoder of the eSBR tools: include/xaac_pvc.h's arithmetic, shared by the
 * HIP kernel (pvc_kernel.hip: a team of 64 lanes per channel) and the CPU oracle (oracle/oracle_pvc.cpp: a team of one).
 * Restates ixheaacd_qmf_enrg_calc (decoder/ixheaacd_sbr_dec.c:80-129) and ixheaacd_pvc_process
 * (decoder/ixheaacd_pred_vec_block.c:30-240) value for value: every float operation is the reference's, in its order (sums
 * of a group's bands ascending, the smoothing window's taps from the newest slot back, the prediction's three low groups in
 * order); only the assignment of independent results to lanes is new.  Double log10 / pow are the C library's on the host and
 * the device library's on the GPU (as in esbr_core.h's pre-flattening: no differing word met so far; tests/test_pvc.py
 * walks reference-made frames on the device).
 */
#ifndef XAAC_PVC_CORE_H
#define XAAC_PVC_CORE_H

#include <math.h>

#include "../../include/xaac_pvc.h"
#include "fx.h"
#include "fx_libm.h"

#if defined(__HIPCC__)
#define XAAC_TAB_QUAL static __device__ const
#include "tables_pvc.inc"
#undef XAAC_TAB_QUAL
#else
#include "tables_pvc.inc"
#endif

#pragma clang fp contract(off)

struct XpCx { /* the team working on one channel: lane of n, sync() between phases that exchange values through w */
  int lane, n;
  FX_MEMBER void sync() const {
#if defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
#endif
  }
};
#define XP_PAR(k, b0, b1) for (int k = (b0) + cx.lane; k < (b1); k += cx.n)

struct XpWork { /* shared between the lanes (LDS on the device) */
  float esg[2 * XAAC_PVC_SLOTS - 1][XAAC_PVC_NB_LOW]; /* ia_pvc_data_struct::esg */
  float fresh[XAAC_PVC_SLOTS][XAAC_PVC_NB_LOW];       /* the frame's own rows before the restart rule */
  float smooth[XAAC_PVC_SLOTS][XAAC_PVC_NB_LOW];      /* smooth_esg_arr */
  float high[XAAC_PVC_SLOTS][8];                      /* 10 ^ (sbr_range_esg_arr / 10) */
};

/* the frame's parameters are inside what the reference's loops and tables cover (the reference itself does not look) */
FX_HD bool xp_frame_ok(const xaac_pvc_frame *f, size_t qmf_stride) {
  if (f->pvc_mode != 1 && f->pvc_mode != 2) return false;       /* pred_vec_block.c:218: returns -1 */
  if (f->pvc_rate != 2 && f->pvc_rate != 4) return false;        /* 8 / pvc_rate, 12 / pvc_rate */
  if (f->first_bnd_idx < 0 || f->first_bnd_idx > 32) return false; /* the low groups end below it, inside a 32-band row */
  /* 4:1: ixheaacd_qmf_enrg_calc (sbr_dec.c:83-107) fills bands 0..15 of a row only -- above them the reference reads what an
     earlier frame left, which is not a function of this frame -- and it reads 64 QMF rows, not 32 */
  if (f->pvc_rate == 4 && (f->first_bnd_idx > 16 || qmf_stride < (size_t)64 * 64)) return false;
  if (f->first_pvc_timeslot < 0 || f->first_pvc_timeslot > XAAC_PVC_SLOTS - 1) return false;
  for (int t = 0; t < XAAC_PVC_SLOTS; t++)
    if (f->pvc_id[t] >= 128) return false;                       /* code book 2 has 128 entries */
  return true;
}

/* energy of one PVC time slot's low band j: ixheaacd_qmf_enrg_calc's two steps for the rows of slot t */
FX_HD float xp_slot_energy(const float *re, const float *im, int rate, int low_power, int t, int j) {
  const int rows = rate == 4 ? 4 : 2;
  float e[4];
  for (int r = 0; r < rows; r++) {
    const int o = 64 * (rows * t + r) + j;
    float v = re[o] * re[o];
    if (!low_power) v += im[o] * im[o];
    e[r] = v;
  }
  return rows == 4 ? (e[0] + e[1] + e[2] + e[3]) * 0.25f : (e[0] + e[1]) * 0.5f;
}

/* One channel's frame.  re / im: row 2 of the channel's QMF buffers; out: [16][64].  Returns 0 or -1 (nothing written). */
FX_HD int xp_process(const XpCx cx, XpWork *w, const xaac_pvc_frame *f, const float *re, const float *im, size_t qmf_stride,
                     xaac_pvc_state *st, float *out) {
  if (!xp_frame_ok(f, qmf_stride)) return -1;
  const int rate = f->pvc_rate, mode1 = f->pvc_mode == 1;
  const int nb_high = mode1 ? 8 : 6, per_grp = (mode1 ? 8 : 12) / rate, lbw = 8 / rate;
  const int nts = mode1 ? (f->ns_mode ? 4 : 16) : (f->ns_mode ? 3 : 12);
  const float *wind = mode1 ? (f->ns_mode ? xaac_pvc_wind_ns4 : xaac_pvc_wind_ns16) : (f->ns_mode ? xaac_pvc_wind_ns3 : xaac_pvc_wind_ns12);
  const float *q = mode1 ? xaac_pvc_q_fac_1 : xaac_pvc_q_fac_2;
  const int8_t *tab1 = mode1 ? xaac_pvc_tab1_1 : xaac_pvc_tab1_2, *tab2 = mode1 ? xaac_pvc_tab2_1 : xaac_pvc_tab2_2;
  const uint8_t *bound = mode1 ? xaac_pvc_id_bound_1 : xaac_pvc_id_bound_2;
  const int first = f->first_bnd_idx, first_slot = f->first_pvc_timeslot;
  const bool restart = st->prev_pvc_flg == 0 || first * rate != st->prev_first_bnd_idx * st->prev_pvc_rate; /* :94-97 */
  /* ixheaacd_pvc_qmf_grouping (:62): the frame's 16 x 3 grouped energies in dB, behind the 15 rows of history */
  XP_PAR(i, 0, XAAC_PVC_SLOTS * XAAC_PVC_NB_LOW) {
    const int t = i / 3, ksg = i % 3, start = first - lbw * XAAC_PVC_NB_LOW + lbw * ksg;
    float esg = 0.1f; /* PVC_ESG_MIN_VAL */
    if (start >= 0) {
      esg = 0.0f;
      for (int ib = start; ib < start + lbw; ib++) esg += xp_slot_energy(re, im, rate, f->low_power, t, ib);
      esg = esg / (float)lbw;
    }
    w->fresh[t][ksg] = esg > 0.1f ? 10 * xm_log10f_of(esg) : -10.0f;
  }
  cx.sync();
  /* rows 0 .. 14: the history -- or, at a restart, like every row in front of the first PVC slot's, that slot's values (:98-104) */
  const int fill = restart ? XAAC_PVC_SLOTS - 1 + first_slot : 0;
  XP_PAR(i, 0, (2 * XAAC_PVC_SLOTS - 1) * XAAC_PVC_NB_LOW) {
    const int r = i / 3, c = i % 3;
    w->esg[r][c] = r < fill ? w->fresh[first_slot][c] : (r < XAAC_PVC_SLOTS - 1 ? st->esg[r][c] : w->fresh[r - (XAAC_PVC_SLOTS - 1)][c]);
  }
  cx.sync();
  /* ixheaacd_pvc_time_smoothing (:109): taps from the slot itself back */
  XP_PAR(i, 0, XAAC_PVC_SLOTS * XAAC_PVC_NB_LOW) {
    const int t = i / 3, ksg = i % 3;
    float acc = 0.0f;
    for (int k = 0; k < nts; k++) acc += w->esg[t + XAAC_PVC_SLOTS - 1 - k][ksg] * wind[k];
    w->smooth[t][ksg] = acc;
  }
  cx.sync();
  /* ixheaacd_pvc_pred_env_sf (:131) and the power of ten of ixheaacd_pvc_sb_parsing (:46) */
  XP_PAR(i, 0, XAAC_PVC_SLOTS * nb_high) {
    const int t = i / nb_high, ksg = i % nb_high, id = f->pvc_id[t];
    const int grp = id < bound[0] ? 0 : (id < bound[1] ? 1 : 2);
    float r = (float)tab2[id * nb_high + ksg] * q[XAAC_PVC_NB_LOW];
    for (int kb = 0; kb < XAAC_PVC_NB_LOW; kb++) {
      const float c = (float)tab1[(grp * XAAC_PVC_NB_LOW + kb) * nb_high + ksg] * q[kb];
      r += c * w->smooth[t][kb];
    }
    w->high[t][ksg] = xm_pow10_tenth(r);
  }
  cx.sync();
  /* ixheaacd_pvc_sb_parsing (:30): group g starts at first + g * per_grp; every group but the first runs on to band 63 when
     its natural end would pass it and the last one always does, later groups overwriting earlier ones -- so a band takes
     the last group that starts at or below it */
  XP_PAR(i, 0, XAAC_PVC_SLOTS * 64) {
    const int t = i >> 6, k = i & 63;
    float v = 0.0f; /* sbr_dec.c:704: the buffer is cleared before every frame */
    if (k >= first) {
      int g = (k - first) / per_grp;
      if (g > nb_high - 1) g = nb_high - 1;
      v = w->high[t][g];
    }
    out[i] = v;
  }
  /* the history shift (:234) and the call site's bookkeeping (sbr_dec.c:945, :951-953; pred_vec_block.c:222) */
  XP_PAR(i, 0, (XAAC_PVC_SLOTS - 1) * XAAC_PVC_NB_LOW) st->esg[i / 3][i % 3] = w->esg[XAAC_PVC_SLOTS + i / 3][i % 3];
  if (cx.lane == 0) {
    st->prev_pvc_id = f->pvc_id[XAAC_PVC_SLOTS - 1];
    st->prev_pvc_flg = 1;
    st->prev_first_bnd_idx = (int16_t)first;
    st->prev_pvc_rate = (uint8_t)rate;
  }
  return 0;
}

#endif /* XAAC_PVC_CORE_H */
