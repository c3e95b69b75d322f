/*
 * oracle/ref_hbe_adapter.c -- TEST INFRASTRUCTURE ONLY.
 * Drives the reference's harmonic-transposer banks -- ixheaacd_real_synth_filt / ixheaacd_complex_anal_filt
 * (decoder/ixheaacd_esbr_polyphase.c:157 / :48) -- on a transposer instance the reference initialises itself
 * (ixheaacd_esbr_hbe_data_init, sbrdec_initfuncs.c:86; ixheaacd_qmf_hbe_data_reinit, hbe_trans.c:102), loaded from and
 * stored back to the boundary struct of include/xaac_hbe.h.  Contains no reference code.
 */
#include "ref_convert.h"
#include "ixheaacd_qmf_poly.h"
#include "xaac_hbe.h"

VOID ixheaacd_esbr_hbe_data_init(ia_esbr_hbe_txposer_struct *pstr_esbr_hbe_txposer, const WORD32 num_aac_samples,
                                 WORD32 samp_fac_4_flag, const WORD32 num_out_samples, VOID *persistent_hbe_mem,
                                 WORD32 *total_persistant);

static __thread ia_esbr_hbe_txposer_struct hbe_t;
static __thread double hbe_mem[(80 * 1024) / 8];

/* a transposer for 1024-sample cores at 2:1, its frequency tables reduced to the two borders reinit reads the bank
   parameters from; returns 0 when the parameters it derives are the ones in st */
static int hbe_load(const xaac_hbe_state *st) {
  WORD32 used = 0;
  int i;
  WORD16 lo[2], hi[2], nsf[2] = {1, 1};
  WORD16 *tab[2];
  ixheaacd_esbr_hbe_data_init(&hbe_t, 1024, 0, 2048, hbe_mem, &used);
  if (used > (WORD32)sizeof(hbe_mem)) return -3;
  lo[0] = hi[0] = (WORD16)st->start_band;
  lo[1] = hi[1] = (WORD16)st->end_band;
  tab[0] = lo;
  tab[1] = hi;
  if (ixheaacd_qmf_hbe_data_reinit(&hbe_t, tab, nsf, 0)) return -1;
  if (hbe_t.synth_size != st->synth_size || hbe_t.k_start != st->k_start) return -2;
  memcpy(hbe_t.ptr_input_buf, st->input_buf, sizeof(st->input_buf));
  memcpy(hbe_t.synth_buf, st->synth_buf, sizeof(st->synth_buf));
  memcpy(hbe_t.analy_buf, st->analy_buf, sizeof(st->analy_buf));
  for (i = 0; i < XAAC_HBE_NO_BINS; i++) memcpy(hbe_t.qmf_in_buf[i], st->qmf_in_buf[i], sizeof(st->qmf_in_buf[i]));
  for (i = 0; i < 2 * XAAC_HBE_NO_BINS; i++) memcpy(hbe_t.qmf_out_buf[i], st->qmf_out_buf[i], sizeof(st->qmf_out_buf[i]));
  for (i = 0; i < 6; i++) hbe_t.x_over_qmf[i] = st->x_over_qmf[i];
  hbe_t.max_stretch = st->max_stretch;
  return 0;
}

static void hbe_store(xaac_hbe_state *st) {
  int i;
  memcpy(st->input_buf, hbe_t.ptr_input_buf, sizeof(st->input_buf));
  memcpy(st->synth_buf, hbe_t.synth_buf, sizeof(st->synth_buf));
  memcpy(st->analy_buf, hbe_t.analy_buf, sizeof(st->analy_buf));
  for (i = 0; i < XAAC_HBE_NO_BINS; i++) memcpy(st->qmf_in_buf[i], hbe_t.qmf_in_buf[i], sizeof(st->qmf_in_buf[i]));
  for (i = 0; i < 2 * XAAC_HBE_NO_BINS; i++) memcpy(st->qmf_out_buf[i], hbe_t.qmf_out_buf[i], sizeof(st->qmf_out_buf[i]));
}

/* what ixheaacd_qmf_hbe_data_reinit derives from an SBR header's frequency tables */
int ref_hbe_reinit(const int16_t *tbl_lo, int n_lo, const int16_t *tbl_hi, int n_hi, xaac_hbe_state *st) {
  WORD32 used = 0;
  int i;
  WORD16 lo[64], hi[64], nsf[2];
  WORD16 *tab[2];
  if (n_lo < 0 || n_lo > 62 || n_hi < 0 || n_hi > 62) return -1;
  ixheaacd_esbr_hbe_data_init(&hbe_t, 1024, 0, 2048, hbe_mem, &used);
  for (i = 0; i <= n_lo; i++) lo[i] = tbl_lo[i];
  for (i = 0; i <= n_hi; i++) hi[i] = tbl_hi[i];
  nsf[0] = (WORD16)n_lo;
  nsf[1] = (WORD16)n_hi;
  tab[0] = lo;
  tab[1] = hi;
  if (ixheaacd_qmf_hbe_data_reinit(&hbe_t, tab, nsf, 0)) return -1;
  st->synth_size = hbe_t.synth_size;
  st->k_start = hbe_t.k_start;
  st->start_band = hbe_t.start_band;
  st->end_band = hbe_t.end_band;
  for (i = 0; i < 6; i++) st->x_over_qmf[i] = hbe_t.x_over_qmf[i];
  st->max_stretch = hbe_t.max_stretch;
  st->fft_ready = 0;
  return 0;
}

int ref_hbe_real_synth(xaac_hbe_state *st, const float *qmf_re, const float *qmf_im, int num_columns) {
  int rc = hbe_load(st);
  if (rc) return rc;
  rc = ixheaacd_real_synth_filt(&hbe_t, num_columns, (FLOAT32(*)[64])qmf_re, (FLOAT32(*)[64])qmf_im);
  hbe_store(st);
  return rc;
}

int ref_hbe_cplx_anal(xaac_hbe_state *st) {
  int rc = hbe_load(st);
  if (rc) return rc;
  rc = ixheaacd_complex_anal_filt(&hbe_t);
  hbe_store(st);
  return rc;
}

void ixheaac_real_synth_fft_p2(FLOAT32 *ptr_x, FLOAT32 *ptr_y, WORD32 npoints);
void ixheaac_cmplx_anal_fft_p2(FLOAT32 *ptr_x, FLOAT32 *ptr_y, WORD32 npoints);

/* ixheaacd_qmf_hbe_apply (hbe_trans.c:224) on the frequency tables the state's parameters were derived from.
   pv_re / pv_im: [32][64].  -2: the tables do not give the state's parameters. */
int ref_hbe_apply(xaac_hbe_state *st, const int16_t *tbl_lo, int n_lo, const int16_t *tbl_hi, int n_hi, const float *qmf_re,
                  const float *qmf_im, int pitch_in_bins, float *pv_re, float *pv_im) {
  static __thread ia_freq_band_data_struct fb;
  static __thread ia_sbr_header_data_struct hd;
  WORD32 used = 0;
  int i, rc;
  WORD16 nsf[2];
  if (n_lo < 0 || n_lo > MAX_FREQ_COEFFS / 2 || n_hi < 0 || n_hi > MAX_FREQ_COEFFS) return -1;
  memset(&fb, 0, sizeof(fb));
  memset(&hd, 0, sizeof(hd));
  for (i = 0; i <= n_lo; i++) fb.freq_band_tbl_lo[i] = tbl_lo[i];
  for (i = 0; i <= n_hi; i++) fb.freq_band_tbl_hi[i] = tbl_hi[i];
  fb.freq_band_table[0] = fb.freq_band_tbl_lo;
  fb.freq_band_table[1] = fb.freq_band_tbl_hi;
  fb.num_sf_bands[0] = (WORD16)n_lo;
  fb.num_sf_bands[1] = (WORD16)n_hi;
  nsf[0] = (WORD16)n_lo;
  nsf[1] = (WORD16)n_hi;
  hd.pstr_freq_band_data = &fb;
  hd.is_usf_4 = 0;
  ixheaacd_esbr_hbe_data_init(&hbe_t, 1024, 0, 2048, hbe_mem, &used);
  hbe_t.max_stretch = st->max_stretch; /* reinit leaves it alone when four patches fit below the end band */
  if (ixheaacd_qmf_hbe_data_reinit(&hbe_t, fb.freq_band_table, nsf, 0)) return -1;
  if (hbe_t.synth_size != st->synth_size || hbe_t.k_start != st->k_start || hbe_t.start_band != st->start_band ||
      hbe_t.end_band != st->end_band || hbe_t.max_stretch != st->max_stretch)
    return -2;
  for (i = 0; i < 6; i++)
    if (hbe_t.x_over_qmf[i] != st->x_over_qmf[i]) return -2;
  memcpy(hbe_t.ptr_input_buf, st->input_buf, sizeof(st->input_buf));
  memcpy(hbe_t.synth_buf, st->synth_buf, sizeof(st->synth_buf));
  memcpy(hbe_t.analy_buf, st->analy_buf, sizeof(st->analy_buf));
  for (i = 0; i < XAAC_HBE_NO_BINS; i++) memcpy(hbe_t.qmf_in_buf[i], st->qmf_in_buf[i], sizeof(st->qmf_in_buf[i]));
  for (i = 0; i < 2 * XAAC_HBE_NO_BINS; i++) memcpy(hbe_t.qmf_out_buf[i], st->qmf_out_buf[i], sizeof(st->qmf_out_buf[i]));
  if (!st->fft_ready) {
    hbe_t.ixheaacd_real_synth_fft = NULL;
    hbe_t.ixheaacd_cmplx_anal_fft = NULL;
  } else if (hbe_t.ixheaacd_cmplx_anal_fft == NULL) { /* size 20 configured after another size: the old pointers stay */
    hbe_t.ixheaacd_real_synth_fft = &ixheaac_real_synth_fft_p2;
    hbe_t.ixheaacd_cmplx_anal_fft = &ixheaac_cmplx_anal_fft_p2;
  }
  rc = ixheaacd_qmf_hbe_apply(&hbe_t, (FLOAT32(*)[64])qmf_re, (FLOAT32(*)[64])qmf_im, XAAC_HBE_NO_BINS, (FLOAT32(*)[64])pv_re,
                              (FLOAT32(*)[64])pv_im, pitch_in_bins, &hd);
  hbe_store(st);
  st->fft_ready = hbe_t.ixheaacd_cmplx_anal_fft != NULL;
  return rc;
}

WORD32 ixheaacd_dft_hbe_cplx_anal_filt(ia_esbr_hbe_txposer_struct *ptr_hbe_txposer, FLOAT32 qmf_buf_real[][64],
                                       FLOAT32 qmf_buf_imag[][64]);
WORD32 ixheaacd_dft_hbe_data_reinit(ia_esbr_hbe_txposer_struct *ptr_hbe_txposer, WORD16 *p_freq_band_tab[2], WORD16 *p_num_sfb);

/* the DFT transposer's analysis bank on a transposer the reference sets up itself from two frequency tables
   (ixheaacd_dft_hbe_data_reinit, hbe_dft_trans.c:272): its coefficient matrices are copied out for the caller
   (coef_re / coef_im [64][128]), the delay line goes in and out through st, time_in -> ptr_output_buf (4096 floats) */
int ref_hbe_dft_anal(const int16_t *tbl_lo, int n_lo, const int16_t *tbl_hi, int n_hi, xaac_hbe_dft_anal_state *st,
                     const float *time_in, int n_time, float *coef_re, float *coef_im, float *qmf_re, float *qmf_im) {
  WORD32 used = 0;
  int i;
  WORD16 lo[64], hi[64], nsf[2];
  WORD16 *tab[2];
  if (n_lo < 0 || n_lo > 62 || n_hi < 0 || n_hi > 62 || n_time > 4096) return -1;
  ixheaacd_esbr_hbe_data_init(&hbe_t, 1024, 0, 2048, hbe_mem, &used);
  for (i = 0; i <= n_lo; i++) lo[i] = tbl_lo[i];
  for (i = 0; i <= n_hi; i++) hi[i] = tbl_hi[i];
  nsf[0] = (WORD16)n_lo;
  nsf[1] = (WORD16)n_hi;
  tab[0] = lo;
  tab[1] = hi;
  if (ixheaacd_dft_hbe_data_reinit(&hbe_t, tab, nsf)) return -1;
  if (st->analy_size == 0) { /* first call: report what the reference derived */
    st->analy_size = hbe_t.analy_size;
    st->a_start = hbe_t.a_start;
  }
  if (hbe_t.analy_size != st->analy_size || hbe_t.a_start != st->a_start) return -2;
  memcpy(coef_re, hbe_t.str_dft_hbe_anal_coeff.real, sizeof(hbe_t.str_dft_hbe_anal_coeff.real));
  memcpy(coef_im, hbe_t.str_dft_hbe_anal_coeff.imag, sizeof(hbe_t.str_dft_hbe_anal_coeff.imag));
  memcpy(hbe_t.analy_buf, st->analy_buf, sizeof(st->analy_buf));
  memset(hbe_t.ptr_output_buf, 0, 4096 * sizeof(FLOAT32));
  memcpy(hbe_t.ptr_output_buf, time_in, sizeof(FLOAT32) * n_time);
  i = ixheaacd_dft_hbe_cplx_anal_filt(&hbe_t, (FLOAT32(*)[64])qmf_re, (FLOAT32(*)[64])qmf_im);
  memcpy(st->analy_buf, hbe_t.analy_buf, sizeof(st->analy_buf));
  return i;
}

WORD32 ixheaacd_dft_hbe_apply(ia_esbr_hbe_txposer_struct *ptr_hbe_txposer, FLOAT32 qmf_buf_real[][64], FLOAT32 qmf_buf_imag[][64],
                              WORD32 num_columns, FLOAT32 pv_qmf_buf_real[][64], FLOAT32 pv_qmf_buf_imag[][64], WORD32 pitch_in_bins,
                              FLOAT32 *dft_hbe_scratch_buf);

/* the DFT transposer the reference sets up from two frequency tables (ixheaacd_dft_hbe_data_reinit): sizes into st, windows
   into cfg, the analysis bank's coefficient matrices into coef_re / coef_im ([64][128]).  max_stretch goes in through st (the
   reference keeps the old value when four patches fit below the end band) and comes back as derived. */
static int hbe_dft_setup(const int16_t *tbl_lo, int n_lo, const int16_t *tbl_hi, int n_hi, int max_stretch_in) {
  WORD32 used = 0;
  int i;
  static __thread WORD16 lo[64], hi[64], nsf[2];
  WORD16 *tab[2];
  if (n_lo < 0 || n_lo > 62 || n_hi < 0 || n_hi > 62) return -1;
  ixheaacd_esbr_hbe_data_init(&hbe_t, 1024, 0, 2048, hbe_mem, &used);
  for (i = 0; i <= n_lo; i++) lo[i] = tbl_lo[i];
  for (i = 0; i <= n_hi; i++) hi[i] = tbl_hi[i];
  nsf[0] = (WORD16)n_lo;
  nsf[1] = (WORD16)n_hi;
  tab[0] = lo;
  tab[1] = hi;
  hbe_t.max_stretch = max_stretch_in;
  return ixheaacd_dft_hbe_data_reinit(&hbe_t, tab, nsf) ? -1 : 0;
}

int ref_hbe_dft_reinit(const int16_t *tbl_lo, int n_lo, const int16_t *tbl_hi, int n_hi, xaac_hbe_dft_state *st, xaac_hbe_dft_cfg *cfg,
                       float *coef_re, float *coef_im) {
  int t, o;
  if (hbe_dft_setup(tbl_lo, n_lo, tbl_hi, n_hi, st->max_stretch)) return -1;
  st->synth_size = hbe_t.synth_size;
  st->k_start = hbe_t.k_start;
  st->start_band = hbe_t.start_band;
  st->end_band = hbe_t.end_band;
  st->max_stretch = hbe_t.max_stretch;
  for (t = 0; t < 6; t++) st->x_over_qmf[t] = hbe_t.x_over_qmf[t];
  st->anal.analy_size = hbe_t.analy_size;
  st->anal.a_start = hbe_t.a_start;
  memset(cfg, 0, sizeof(*cfg));
  if (hbe_t.ana_fft_size[0] > XAAC_HBE_DFT_MAX_ANA || hbe_t.syn_fft_size[0] > XAAC_HBE_DFT_MAX_SYN) return -4;
  memcpy(cfg->anal_window, hbe_t.analysis_window_buf, sizeof(float) * hbe_t.ana_fft_size[0]);
  memcpy(cfg->synth_window, hbe_t.synthesis_window_buf, sizeof(float) * hbe_t.syn_fft_size[0]);
  for (t = 0; t < 3; t++)
    for (o = 0; o < 2; o++) memcpy(cfg->fd_win[t][o], hbe_t.fd_win_buf[t][o], sizeof(cfg->fd_win[t][o]));
  memcpy(coef_re, hbe_t.str_dft_hbe_anal_coeff.real, sizeof(hbe_t.str_dft_hbe_anal_coeff.real));
  memcpy(coef_im, hbe_t.str_dft_hbe_anal_coeff.imag, sizeof(hbe_t.str_dft_hbe_anal_coeff.imag));
  return 0;
}

/* ixheaacd_dft_hbe_apply on a transposer set up from the tables, with st's signals and delay lines; pv_re / pv_im: [34][64]
   in/out.  -2: the tables do not give st's sizes. */
int ref_hbe_dft_apply(const int16_t *tbl_lo, int n_lo, const int16_t *tbl_hi, int n_hi, int max_stretch_before, xaac_hbe_dft_state *st,
                      const float *qmf_re, const float *qmf_im, int pitch_in_bins, int oversampling, float *pv_re, float *pv_im) {
  static __thread FLOAT32 scratch[16384];
  int rc;
  if (hbe_dft_setup(tbl_lo, n_lo, tbl_hi, n_hi, max_stretch_before)) return -1;
  if (hbe_t.synth_size != st->synth_size || hbe_t.k_start != st->k_start || hbe_t.max_stretch != st->max_stretch ||
      hbe_t.analy_size != st->anal.analy_size || hbe_t.a_start != st->anal.a_start)
    return -2;
  memcpy(hbe_t.ptr_input_buf, st->input_buf, sizeof(st->input_buf));
  memcpy(hbe_t.output_buf, st->output_buf, sizeof(st->output_buf));
  memcpy(hbe_t.synth_buf, st->synth_buf, sizeof(st->synth_buf));
  memcpy(hbe_t.analy_buf, st->anal.analy_buf, sizeof(st->anal.analy_buf));
  hbe_t.oversampling_flag = oversampling ? 1 : 0;
  rc = ixheaacd_dft_hbe_apply(&hbe_t, (FLOAT32(*)[64])qmf_re, (FLOAT32(*)[64])qmf_im, XAAC_HBE_NO_BINS, (FLOAT32(*)[64])pv_re,
                              (FLOAT32(*)[64])pv_im, pitch_in_bins, scratch);
  memcpy(st->input_buf, hbe_t.ptr_input_buf, sizeof(st->input_buf));
  memcpy(st->output_buf, hbe_t.output_buf, sizeof(st->output_buf));
  memcpy(st->synth_buf, hbe_t.synth_buf, sizeof(st->synth_buf));
  memcpy(st->anal.analy_buf, hbe_t.analy_buf, sizeof(st->anal.analy_buf));
  return rc;
}
