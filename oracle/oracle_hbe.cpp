/*
 * oracle_hbe.cpp -- TEST INFRASTRUCTURE: the sequential CPU run of libxaac_amd/csrc/hbe_poly.h (the harmonic
 * transposer's polyphase banks, decoder/ixheaacd_esbr_polyphase.c:48-274).  Pinned bit for bit against the compiled
 * reference by tests/test_hbe_oracle_vs_reference.py (oracle/ref_hbe_adapter.c).  Only tests/, __graft_entry__.smoke()
 * and bench.py's checker legs may use it.
 */
#include <string.h>

#include "../libxaac_amd/csrc/hbe_poly.h"

extern "C" {

/* ixheaacd_real_synth_filt (esbr_polyphase.c:157), esbr_hq = 0.  Returns 0, or -1 for parameters outside the tables. */
int xo_hbe_real_synth(xaac_hbe_state *st, const float *qmf_re, const float *qmf_im, int num_columns) {
  const int s = st->synth_size, ks = st->k_start;
  if (!xh_size_ok(s) || ks < 0 || ks + s > 64 || ks * 32 + 2 * s > 7 * 64 || num_columns < 0 || num_columns > 32) return -1;
  static thread_local float v[32 + 9][40], w[XH_FFT_SCRATCH];
  for (int c = -9; c < 0; c++)
    for (int t = 0; t < 2 * s; t++) v[c + 9][t] = xh_synth_hist(st->synth_buf, s, c, t);
  for (int idx = 0; idx < num_columns; idx++) xh_synth_column(qmf_re + 64 * idx, qmf_im + 64 * idx, s, ks, v[idx + 9], w);
  const auto vv = [&](int c, int t) { return v[c + 9][t]; };
  for (int idx = 0; idx < num_columns; idx++)
    for (int i = 0; i < s; i++) st->input_buf[(idx + 1) * s + i] = xh_synth_out(vv, s, idx, i);
  float nb[400];
  for (int m = 0; m < 10; m++)
    for (int t = 0; t < 2 * s; t++) nb[2 * s * m + t] = v[num_columns - 1 - m + 9][t];
  memcpy(st->synth_buf, nb, sizeof(float) * 20 * s);
  return 0;
}

/* ixheaacd_complex_anal_filt (esbr_polyphase.c:48), esbr_hq = 0: no_bins / 2 = 16 columns into qmf_in_buf rows 12..27 */
int xo_hbe_cplx_anal(xaac_hbe_state *st) {
  const int s = st->synth_size, ks = st->k_start, a = 2 * s;
  if (!xh_size_ok(s) || ks < 0 || 4 * ks + 2 * a > 128) return -1;
  static thread_local float u[160], w[XH_FFT_SCRATCH];
  for (int idx = 0; idx < XAAC_HBE_NO_BINS / 2; idx++) {
    for (int i = 0; i < 2 * a; i++) u[i] = xh_anal_u(st->input_buf, st->analy_buf, a, idx, i);
    float *row = st->qmf_in_buf[idx + XAAC_HBE_OPER_WIN_LEN - 1];
    memset(row, 0, sizeof(float) * 128);
    xh_anal_column(u, a, row + 4 * ks, w);
  }
  float nb[400];
  for (int n = 0; n < 10 * a; n++) nb[n] = xh_anal_x(st->input_buf, st->analy_buf, a, XAAC_HBE_NO_BINS / 2 - 1, n);
  memcpy(st->analy_buf, nb, sizeof(float) * 10 * a);
  return 0;
}

}  // extern "C"
