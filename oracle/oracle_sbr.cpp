/*
 * oracle/oracle_sbr.cpp -- TEST INFRASTRUCTURE ONLY (checker + CPU baseline).
 *
 * One channel-frame of the fixed-point low-power SBR decoder: the restatement of
 * ixheaacd_sbr_dec (decoder/ixheaacd_sbr_dec.c:662, Path B, low_pow_flag = 1, HE-AACv1) on the
 * boundary formats of include/xaac_sbr.h.  Arithmetic comes from libxaac_amd/csrc/sbr_core.h and
 * sbr_qmf.h (shared with the GPU); the two QMF banks are the oracle's own ring-faithful versions
 * (oracle_qmf.cpp).
 *
 * Parity status: PINNED on captured frames -- tests/test_sbr_oracle_vs_capture.py replays records
 * that oracle/ref_capture.c wrote while the compiled reference decoded real HE-AAC streams
 * (inputs + state before -> PCM + state after must match word for word).
 */
#include <stdint.h>
#include <string.h>
#ifdef XO_PS_DEBUG
#include <stdio.h>
#include <stdlib.h>
#endif

#include "../include/xaac_amd.h" /* xaac_sbr_eld_state */
#include "../libxaac_amd/csrc/sbr_core.h"
#include "../libxaac_amd/csrc/sbr_ps.h"
#include "../libxaac_amd/csrc/sbr_ps_frame.h"
#include "oracle_qmf.h"

static const int16_t *rand_hi_table() {
  static const struct Tab {
    int16_t v[568];
    Tab() {
      for (int i = 0; i < 568; i++) v[i] = (int16_t)(xaac_sbr_rand_ph[i] >> 16);
    }
  } tab;
  return tab.v;
}

/* the QMF matrices: static per thread (the callers are test loops), or -- XO_MATRIX_ON_STACK, tests/test_sbr_core_sanitized.py --
   on the stack, where AddressSanitizer sees a write in front of or behind them */
#ifdef XO_MATRIX_ON_STACK
#define XO_MATRIX
#else
#define XO_MATRIX static thread_local
#endif
/* down-sampled synthesis bank (32 channels) for the call in flight: set by the *_ds entry points */
static thread_local int g_ds = 0;
/* 1: run the parametric-stereo tool through the product's frame-at-once arrangement (sbr_ps_frame.h, lane count 1)
   instead of the slot loop below -- tests/test_ps_frame_cpu.py checks the two against each other on the host */
static thread_local int g_ps_phased = 0;

extern "C" int xo_sbr_dec_lp(const xaac_sbr_header *h, const xaac_sbr_frame *f, xaac_sbr_state *st,
                             const int16_t *pcm_in, int in_stride, int16_t *pcm_out, int out_stride) {
  XO_MATRIX int32_t buf[40 * 64];
  XsQmf x = {buf};
  const XsCx cx = {0, 1}; /* sequential execution of the shared core */
  XsWork w;
  memset(buf, 0, sizeof(buf));
  /* sbr_dec.c:753: the six overlap slots */
  for (int l = 0; l < 6; l++)
    for (int k = 0; k < 64; k++) x(l, k) = st->overlap[64 * l + k];
  if (xs_side_info_bad(cx, h, f, st)) return -1; /* refused before anything is touched, like sbr_dec.c:733-748 */
  st->lb_scale = 0;
  if (f->apply_processing) xs_rescale_x_overlap(cx, h, f, st, x);
  /* sbr_dec.c:1025: analysis bank into slots 6..37 */
  {
    xo_qmf_ana_state a;
    memcpy(a.ring, st->ana_ring, sizeof(a.ring));
    a.wr = st->ana_wr;
    a.phase = st->ana_phase;
    xo_qmf_analysis(pcm_in, in_stride, &a, 1, st->codec_usb, &x(6, 0), 64);
    memcpy(st->ana_ring, a.ring, sizeof(a.ring));
    st->ana_wr = a.wr;
    st->ana_phase = a.phase;
    st->st_lb_scale = 0;
    st->lb_scale = -10;
  }
  int save_lb_scale = 0;
  if (xs_sbr_core(cx, h, f, f->int_env_sf_arr, f->int_noise_floor, st, x, &w, rand_hi_table(), &save_lb_scale)) return -1;
  /* sbr_dec.c:1273: synthesis bank over slots 0..31 */
  {
    xo_qmf_syn_state s;
    memcpy(s.ring, st->syn_ring, sizeof(s.ring));
    s.drc_offset = st->syn_drc_offset;
    s.phase = st->syn_phase;
    const int16_t sf[4] = {st->lb_scale, st->ov_lb_scale, st->hb_scale, st->st_syn_scale};
    xo_qmf_synthesis_n(&x(0, 0), 64, sf, st->syn_lsb, st->syn_usb, 6, &s, 1, pcm_out, out_stride, g_ds);
    memcpy(st->syn_ring, s.ring, sizeof(s.ring));
    st->syn_drc_offset = s.drc_offset;
    st->syn_phase = s.phase;
  }
  /* sbr_dec.c:1283-1308 */
  for (int l = 0; l < 6; l++)
    for (int k = 0; k < 64; k++) st->overlap[64 * l + k] = x(32 + l, k);
  st->ov_lb_scale = (int16_t)save_lb_scale;
  return 0;
}

static inline int32_t adj_word(int32_t v, int shift) { /* env_calc.c:1099 on one word */
  if (shift == 0) return v;
  if (shift > 31) shift = 31;
  if (shift < -31) shift = -31;
  return shift > 0 ? fx_shlw(v, shift) : (v >> -shift);
}

/* One frame of ixheaacd_sbr_dec in HQ (complex) mode: low_pow_flag = 0, i.e. HE-AAC mono and, with
   pf / ps given and channel_mode = PS_STEREO, HE-AACv2 (sbr_dec.c:1246-1281: the left synthesis bank
   runs the parametric-stereo tool slot by slot, the right one consumes what it leaves in the matrix).
   pcm_out: left at [n * out_stride], right at [n * out_stride + 1]. */
extern "C" int xo_sbr_dec_hq(const xaac_sbr_header *h, const xaac_sbr_frame *f, xaac_sbr_state *st,
                             const xaac_ps_frame *pf, xaac_ps_state *ps, const int16_t *pcm_in, int in_stride,
                             int16_t *pcm_out, int out_stride) {
  XO_MATRIX int32_t buf[41 * 128];
  XsQmfHq x = {buf};
  const XsCx cx = {0, 1};
  XsWork w;
  memset(buf, 0, sizeof(buf));
  /* sbr_dec.c:753: twelve half-rows = six complex overlap slots */
  memcpy(&x(0, 0), st->overlap, sizeof(int32_t) * 12 * 64);
  if (xs_side_info_bad(cx, h, f, st)) return -1; /* refused before anything is touched, like sbr_dec.c:733-748 */
  st->lb_scale = 0;
  if (f->apply_processing) xs_rescale_x_overlap(cx, h, f, st, x);
  {
    xo_qmf_ana_state a;
    memcpy(a.ring, st->ana_ring, sizeof(a.ring));
    a.wr = st->ana_wr;
    a.phase = st->ana_phase;
    xo_qmf_analysis(pcm_in, in_stride, &a, 0, st->codec_usb, &x(6, 0), 128);
    memcpy(st->ana_ring, a.ring, sizeof(a.ring));
    st->ana_wr = a.wr;
    st->ana_phase = a.phase;
    st->st_lb_scale = 0;
    st->lb_scale = -8; /* generic:631 */
  }
  int save_lb_scale = 0;
  if (xs_sbr_core(cx, h, f, f->int_env_sf_arr, f->int_noise_floor, st, x, &w, rand_hi_table(), &save_lb_scale)) return -1;
  xo_qmf_syn_state s;
  memcpy(s.ring, st->syn_ring, sizeof(s.ring));
  s.drc_offset = st->syn_drc_offset;
  s.phase = st->syn_phase;
  int ps_clamped = 0;
  xaac_ps_frame pf_clean;
  if (f->apply_processing && h->channel_mode == 3 && pf && ps && g_ps_phased) {
    pf_clean = *pf;
    ps_clamped = xp_frame_sanitize(cx, &pf_clean);
    static thread_local XpFrameWork wk;
    XO_MATRIX int32_t xr[32 * 128];
    const int st_syn = st->st_syn_scale;
    const int ps_scale = xp_ps_frame(cx, &xaac_ps_tables, ps, &pf_clean, &wk, &x(0, 0), xr, st->lb_scale, st->ov_lb_scale,
                                     st->hb_scale, st_syn, st->syn_lsb, st->syn_usb);
    st->ps_scale = (int16_t)ps_scale;
    const int16_t ready = (int16_t)(st_syn - 8); /* the left rows are already in the bank's scale: its rescale is a no-op */
    const int16_t sf_l[4] = {ready, ready, ready, (int16_t)st_syn};
    xo_qmf_synthesis_n(&x(0, 0), 128, sf_l, st->syn_lsb, st->syn_usb, 6, &s, 0, pcm_out, out_stride, g_ds);
    ps->lb_scale_r = ps->ov_lb_scale_r = ps->hb_scale_r = (int16_t)ps_scale;
    xo_qmf_syn_state r;
    memcpy(r.ring, ps->syn_ring_r, sizeof(r.ring));
    r.drc_offset = ps->syn_drc_offset_r;
    r.phase = ps->syn_phase_r;
    const int16_t sf_r[4] = {(int16_t)ps_scale, (int16_t)ps_scale, (int16_t)ps_scale, ps->st_syn_scale_r};
    xo_qmf_synthesis_n(xr, 128, sf_r, ps->syn_lsb_r, ps->syn_usb_r, 6, &r, 0, pcm_out + 1, out_stride, g_ds);
    memcpy(ps->syn_ring_r, r.ring, sizeof(r.ring));
    ps->syn_drc_offset_r = r.drc_offset;
    ps->syn_phase_r = r.phase;
  } else if (f->apply_processing && h->channel_mode == 3 && pf && ps) {
    pf_clean = *pf;
    ps_clamped = xp_frame_sanitize(cx, &pf_clean);
    pf = &pf_clean;
    const int ps_scale = xp_init_ps_scale(cx, ps, st->lb_scale, st->ov_lb_scale, st->hb_scale);
    st->ps_scale = (int16_t)ps_scale;
    const int lsb = st->syn_lsb, usb = st->syn_usb, st_syn = st->st_syn_scale;
    const int ov_lb_shift = ps_scale - st->ov_lb_scale, lb_shift = ps_scale - st->lb_scale,
              hb_shift = ps_scale - st->hb_scale, common_shift = (st_syn - ps_scale) - 8;
    for (int l = 0; l < 32; l++)
      for (int k = 0; k < 64; k++) {
        const int sh = k < lsb ? (l < 6 ? ov_lb_shift : lb_shift) : (k < usb ? hb_shift : 0);
        x(l, k) = adj_word(x(l, k), sh);
        x.im(l, k) = adj_word(x.im(l, k), sh);
      }
    int env = 0;
    for (int l = 0; l < 32; l++) {
      int32_t right[128];
      XpHyb hy;
      memset(&hy, 0, sizeof(hy));
      int16_t ratio[21];
      int32_t band_pw[64];
      if (env <= XAAC_PS_MAX_ENV && l == pf->border_position[env]) {
        xp_init_rot_env(cx, &xaac_ps_tables, ps, pf, env, usb);
        env++;
      }
      const int shiftdelay = l < 32 - 6 ? 0 : (int16_t)(st->lb_scale - ps_scale); /* thumb_ps_dec.c:77 */
      xp_hybrid_analysis(cx, &xaac_ps_tables, &x(l + 6, 0), &x.im(l + 6, 0), ps, &hy, shiftdelay);
      xp_decorrelation(cx, &xaac_ps_tables, ps, &hy, &x(l, 0), right, ratio, band_pw);
      xp_apply_rot(cx, &xaac_ps_tables, ps, &hy, &x(l, 0), right);
#ifdef XO_PS_DEBUG
      {
        static FILE *df;
        const char *dp = getenv("XO_PS_DUMP");
        if (dp) {
          int32_t sl = l;
          if (!df) df = fopen(dp, "wb");
          fwrite(&sl, 4, 1, df);
          fwrite(&x(l, 0), 4, 128, df);
          fwrite(right, 4, 128, df);
          fflush(df);
        }
      }
#endif
      if (common_shift) /* generic:1610 */
        for (int k = 0; k < 128; k++) {
          int32_t *p = &x(l, 0) + k;
          *p = common_shift < 0 ? fx_shr(*p, -common_shift > 31 ? 31 : -common_shift) : fx_shl_sat(*p, common_shift);
        }
      xo_qmf_synthesis_slot_n(&x(l, 0), &s, l, 0, -(st_syn - 3), pcm_out + (size_t)out_stride * (g_ds ? 32 : 64) * l, out_stride, g_ds);
      memcpy(&x(l, 0), right, sizeof(right));
    }
    /* right channel: all three scales are ps_scale (sbr_dec.c:1261-1264) */
    ps->lb_scale_r = ps->ov_lb_scale_r = ps->hb_scale_r = (int16_t)ps_scale;
    xo_qmf_syn_state r;
    memcpy(r.ring, ps->syn_ring_r, sizeof(r.ring));
    r.drc_offset = ps->syn_drc_offset_r;
    r.phase = ps->syn_phase_r;
    const int16_t sf_r[4] = {(int16_t)ps_scale, (int16_t)ps_scale, (int16_t)ps_scale, ps->st_syn_scale_r};
    xo_qmf_synthesis_n(&x(0, 0), 128, sf_r, ps->syn_lsb_r, ps->syn_usb_r, 6, &r, 0, pcm_out + 1, out_stride, g_ds);
    memcpy(ps->syn_ring_r, r.ring, sizeof(r.ring));
    ps->syn_drc_offset_r = r.drc_offset;
    ps->syn_phase_r = r.phase;
  } else {
    const int16_t sf[4] = {st->lb_scale, st->ov_lb_scale, st->hb_scale, st->st_syn_scale};
    xo_qmf_synthesis_n(&x(0, 0), 128, sf, st->syn_lsb, st->syn_usb, 6, &s, 0, pcm_out, out_stride, g_ds);
  }
  memcpy(st->syn_ring, s.ring, sizeof(s.ring));
  st->syn_drc_offset = s.drc_offset;
  st->syn_phase = s.phase;
  /* sbr_dec.c:1283-1291 copies 6 * 64 words whatever the mode: in HQ that is the first three of the six
     complex overlap slots -- the other three keep what they held (kept as is: it is the reference's output) */
  memcpy(st->overlap, &x(32, 0), sizeof(int32_t) * 6 * 64);
  st->ov_lb_scale = (int16_t)save_lb_scale;
  return ps_clamped ? -1 : 0;
}

/* n independent channel-frames in a C loop (CPU baseline "port" when oracle/_ref did not travel) */
extern "C" int xo_sbr_dec_lp_batch(int n, const xaac_sbr_header *h, const xaac_sbr_frame *f, xaac_sbr_state *st,
                                   const int16_t *pcm_in, int16_t *pcm_out) {
  int bad = 0;
  for (int i = 0; i < n; i++)
    bad += xo_sbr_dec_lp(h + i, f + i, st + i, pcm_in + 1024 * (size_t)i, 1, pcm_out + 2048 * (size_t)i, 1) != 0;
  return bad;
}
/* ixheaacd_sbr_dec for an AAC-ELD channel (low-delay SBR: sbr_dec.c:706-775, :1025-1308 with AOT_ER_AAC_ELD, low_pow_flag 0): no
   overlap slots, n = num_time_slots (16 or 15) QMF slots a frame, the LD banks.  pcm_in: 32 n samples, pcm_out: 64 n.
   handed_on (optional, [n][128]): the region-rescaled rows the synthesis bank leaves / hands on (qmf_dec.c:937-976). */
extern "C" {
void xo_qmf_analysis_eld(const int16_t *pcm, int stride, int16_t *ring, int16_t *state4, int n_slots, int usb, int32_t *qmf,
                         int slot_stride);
void xo_qmf_eld_region_scale(const int32_t *qmf, int slot_stride, const int16_t *sf, int lsb, int usb, int split, int n_slots,
                             int32_t *out);
void xo_qmf_synthesis_eld(const int32_t *qmf, int slot_stride, const int16_t *sf, int lsb, int usb, int split, int16_t *ring,
                          int16_t *state4, int n_slots, int16_t *pcm, int stride);
}
extern "C" int xo_sbr_dec_eld(const xaac_sbr_header *h, const xaac_sbr_frame *f, xaac_sbr_eld_state *st, const int16_t *pcm_in,
                              int in_stride, int16_t *pcm_out, int out_stride, int32_t *handed_on) {
  XO_MATRIX int32_t buf[(2 + 16) * 128];
  XsQmfT<1, 64, 1> x = {buf};
  const XsCx cx = {0, 1};
  XsWork w;
  memset(buf, 0, sizeof(buf));
  if (xs_side_info_bad(cx, h, f, st, 1)) return -1;
  const int n = h->num_time_slots;
  st->lb_scale = 0;                                                   /* sbr_dec.c:767 */
  if (f->apply_processing) xs_rescale_x_overlap(cx, h, f, st, x);     /* its two state words and scale bookkeeping: no rows */
  {
    int16_t s4[4] = {st->ana.wr, st->ana.f1, st->ana.f2, st->ana.fp};
    xo_qmf_analysis_eld(pcm_in, in_stride, st->ana.ring, s4, n, st->codec_usb, &x(0, 0), 128);
    st->ana.wr = s4[0], st->ana.f1 = s4[1], st->ana.f2 = s4[2], st->ana.fp = s4[3];
    st->st_lb_scale = 0;
    st->lb_scale = -9; /* generic:630-636 */
  }
  int save_lb_scale = 0;
  if (xs_sbr_core(cx, h, f, f->int_env_sf_arr, f->int_noise_floor, st, x, &w, rand_hi_table(), &save_lb_scale)) return -1;
  {
    const int16_t sf[4] = {st->lb_scale, st->ov_lb_scale, st->hb_scale, st->st_syn_scale};
    int16_t s4[4] = {st->syn.drc_offset, st->syn.phase, st->syn.fp, st->syn.sixty4};
    if (handed_on) xo_qmf_eld_region_scale(&x(0, 0), 128, sf, st->syn_lsb, st->syn_usb, 0, n, handed_on);
    xo_qmf_synthesis_eld(&x(0, 0), 128, sf, st->syn_lsb, st->syn_usb, 0, st->syn.ring, s4, n, pcm_out, out_stride);
    st->syn.drc_offset = s4[0], st->syn.phase = s4[1], st->syn.fp = s4[2], st->syn.sixty4 = s4[3];
  }
  st->ov_lb_scale = (int16_t)save_lb_scale; /* sbr_dec.c:1304 */
  return 0;
}

extern "C" int xo_sbr_dec_hq_batch(int n, const xaac_sbr_header *h, const xaac_sbr_frame *f, xaac_sbr_state *st,
                                   const xaac_ps_frame *pf, xaac_ps_state *ps, const int16_t *pcm_in, int16_t *pcm_out) {
  int bad = 0;
  for (int i = 0; i < n; i++)
    bad += xo_sbr_dec_hq(h + i, f + i, st + i, pf ? pf + i : nullptr, ps ? ps + i : nullptr, pcm_in + 1024 * (size_t)i, 1,
                         pcm_out + (pf ? 4096 : 2048) * (size_t)i, pf ? 2 : 1) != 0;
  return bad;
}

/* xo_sbr_dec_hq with the parametric-stereo tool run a frame at a time (the product's arrangement for the GPU) */
extern "C" int xo_sbr_dec_hq_phased(const xaac_sbr_header *h, const xaac_sbr_frame *f, xaac_sbr_state *st,
                                    const xaac_ps_frame *pf, xaac_ps_state *ps, const int16_t *pcm_in, int in_stride,
                                    int16_t *pcm_out, int out_stride) {
  g_ps_phased = 1;
  const int rc = xo_sbr_dec_hq(h, f, st, pf, ps, pcm_in, in_stride, pcm_out, out_stride);
  g_ps_phased = 0;
  return rc;
}

/* the same two calls with the down-sampled synthesis bank (1024 output samples per channel) */
extern "C" int xo_sbr_dec_lp_ds(const xaac_sbr_header *h, const xaac_sbr_frame *f, xaac_sbr_state *st,
                                const int16_t *pcm_in, int in_stride, int16_t *pcm_out, int out_stride, int ds) {
  g_ds = ds;
  const int rc = xo_sbr_dec_lp(h, f, st, pcm_in, in_stride, pcm_out, out_stride);
  g_ds = 0;
  return rc;
}
extern "C" int xo_sbr_dec_hq_ds(const xaac_sbr_header *h, const xaac_sbr_frame *f, xaac_sbr_state *st,
                                const xaac_ps_frame *pf, xaac_ps_state *ps, const int16_t *pcm_in, int in_stride,
                                int16_t *pcm_out, int out_stride, int ds) {
  g_ds = ds;
  const int rc = xo_sbr_dec_hq(h, f, st, pf, ps, pcm_in, in_stride, pcm_out, out_stride);
  g_ds = 0;
  return rc;
}
