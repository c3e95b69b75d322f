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

#include "../libxaac_amd/csrc/sbr_core.h"
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

extern "C" int xo_sbr_dec_lp(const xaac_sbr_header *h, const xaac_sbr_frame *f, xaac_sbr_state *st,
                             const int16_t *pcm_in, int in_stride, int16_t *pcm_out, int out_stride) {
  static thread_local int32_t buf[40 * 64];
  XsQmf x = {buf};
  const XsCx cx = {0, 1}; /* sequential execution of the shared core */
  XsWork w;
  memset(buf, 0, sizeof(buf));
  /* sbr_dec.c:753: the six overlap slots */
  for (int l = 0; l < 6; l++)
    for (int k = 0; k < 64; k++) x(l, k) = st->overlap[64 * l + k];
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
  if (xs_sbr_core_lp(cx, h, f, st, x, &w, rand_hi_table(), &save_lb_scale)) return -1;
  /* sbr_dec.c:1273: synthesis bank over slots 0..31 */
  {
    xo_qmf_syn_state s;
    memcpy(s.ring, st->syn_ring, sizeof(s.ring));
    s.drc_offset = st->syn_drc_offset;
    s.phase = st->syn_phase;
    const int16_t sf[4] = {st->lb_scale, st->ov_lb_scale, st->hb_scale, st->st_syn_scale};
    xo_qmf_synthesis(&x(0, 0), 64, sf, st->syn_lsb, st->syn_usb, 6, &s, 1, pcm_out, out_stride);
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
