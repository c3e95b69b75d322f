/*
 * oracle/ref_esbr_adapter.c -- TEST INFRASTRUCTURE ONLY.
 * Drives the real float HF generator and envelope adjuster of the reference's default SBR path --
 * ixheaacd_generate_hf (decoder/ixheaacd_sbrdec_lpfuncs.c:981) and ixheaacd_sbr_env_calc (ixheaacd_esbr_envcal.c:71) --
 * from the boundary structs of include/xaac_esbr.h: fills the reference's own header / frame structs field by field
 * (ref_convert.h's includes give their definitions), calls the reference's symbols on the caller's [rows][64] buffers,
 * and copies the persistent members back.  Contains no reference code.
 */
#include "ref_convert.h"
#include "xaac_esbr.h"

/* ph_re / ph_im: the transposer's rows ([40][64], row 2 = the reference's + SBR_HF_ADJ_OFFSET) and its cross-over bands,
   or NULL (then hbe_flag is 0 as for a USAC stream without harmonic SBR) */
int ref_esbr_hf_env_h(const xaac_sbr_header *h, const xaac_sbr_frame *f, const xaac_esbr_side *sd, xaac_esbr_state *st,
                      float *qmf_re, float *qmf_im, float *out_re, float *out_im, float *ph_re, float *ph_im,
                      const int32_t *x_over_qmf) {
  static __thread ia_freq_band_data_struct fb;
  static __thread ia_sbr_header_data_struct hd;
  static __thread ia_sbr_frame_info_data_struct fd;
  static __thread FLOAT32 scratch[1024], env_out[1024];
  ia_frame_info_struct *fi = &fd.str_frame_info_details;
  int i, rc;
  memset(&fb, 0, sizeof(fb));
  memset(&hd, 0, sizeof(hd));
  memset(&fd, 0, sizeof(fd));
  memset(env_out, 0, sizeof(env_out));
  fb.num_sf_bands[0] = h->num_sf_bands[0];
  fb.num_sf_bands[1] = h->num_sf_bands[1];
  fb.num_nf_bands = h->num_nf_bands;
  fb.num_mf_bands = sd->num_mf_bands;
  fb.sub_band_start = h->sub_band_start;
  fb.sub_band_end = h->sub_band_end;
  fb.num_lf_bands = h->num_lf_bands;
  fb.num_if_bands = h->num_if_bands;
  memcpy(fb.freq_band_tbl_lo, h->freq_band_tbl_lo, sizeof(fb.freq_band_tbl_lo));
  memcpy(fb.freq_band_tbl_hi, h->freq_band_tbl_hi, sizeof(fb.freq_band_tbl_hi));
  memcpy(fb.freq_band_tbl_noise, h->freq_band_tbl_noise, sizeof(fb.freq_band_tbl_noise));
  memcpy(fb.f_master_tbl, sd->f_master_tbl, sizeof(fb.f_master_tbl));
  fb.freq_band_table[0] = fb.freq_band_tbl_lo;
  fb.freq_band_table[1] = fb.freq_band_tbl_hi;
  fb.qmf_sb_prev = sd->qmf_sb_prev;
  hd.pstr_freq_band_data = &fb;
  hd.out_sampling_freq = sd->out_sampling_freq;
  hd.limiter_bands = sd->limiter_bands;
  hd.limiter_gains = h->limiter_gains;
  hd.interpol_freq = h->interpol_freq;
  hd.smoothing_mode = h->smoothing_mode;
  hd.num_time_slots = 16;
  hd.time_step = 2;
  hd.esbr_start_up = st->esbr_start_up;
  hd.esbr_start_up_pvc = st->esbr_start_up;
  hd.enh_sbr = 1;
  fd.pstr_sbr_header = &hd;
  fi->frame_class = f->frame_class;
  fi->num_env = f->num_env;
  fi->transient_env = f->transient_env;
  fi->num_noise_env = f->num_noise_env;
  for (i = 0; i <= XAAC_SBR_MAX_ENVELOPES; i++) fi->border_vec[i] = f->border_vec[i];
  for (i = 0; i < XAAC_SBR_MAX_ENVELOPES; i++) fi->freq_res[i] = f->freq_res[i];
  for (i = 0; i <= XAAC_SBR_MAX_NOISE_ENVELOPES; i++) fi->noise_border_vec[i] = f->noise_border_vec[i];
  for (i = 0; i < XAAC_SBR_MAX_NOISE_VALUES; i++) {
    fd.sbr_invf_mode[i] = f->sbr_invf_mode[i];
    fd.sbr_invf_mode_prev[i] = sd->sbr_invf_mode_prev[i];
    fd.flt_noise_floor[i] = sd->flt_noise_floor[i];
  }
  for (i = 0; i < XAAC_SBR_MAX_FREQ_COEFFS; i++) fd.add_harmonics[i] = f->add_harmonics[i];
  for (i = 0; i < XAAC_SBR_MAX_ENV_VALUES; i++) fd.flt_env_sf_arr[i] = sd->flt_env_sf_arr[i];
  for (i = 0; i < XAAC_SBR_MAX_ENVELOPES; i++) fd.inter_temp_shape_mode[i] = sd->inter_temp_shape_mode[i];
  fd.env_short_flag_prev = st->env_short_flag_prev;
  fd.sbr_patching_mode = (sd->harmonic_sbr & XAAC_ESBR_HARMONIC) ? 0 : 1;
  hd.pre_proc_flag = (sd->harmonic_sbr & XAAC_ESBR_PRE_FLATTEN) != 0;
  fd.prev_sbr_patching_mode = st->prev_sbr_patching_mode;
  fd.pitch_in_bins = sd->pitch_in_bins;
  hd.hbe_flag = ph_re != NULL;
  fd.sbr_mode = ORIG_SBR;
  fd.prev_sbr_mode = ORIG_SBR;
  fd.reset_flag = sd->reset_flag;
  fd.harm_index = st->harm_index;
  fd.phase_index = st->phase_index;
  memcpy(fd.bw_array_prev, st->bw_array_prev, sizeof(fd.bw_array_prev));
  memcpy(fd.harm_flag_prev, st->harm_flag_prev, sizeof(fd.harm_flag_prev));
  memcpy(fd.e_gain, st->e_gain, sizeof(fd.e_gain));
  memcpy(fd.noise_buf, st->noise_buf, sizeof(fd.noise_buf));
  memcpy(fd.lim_table, st->lim_table, sizeof(fd.lim_table));
  memcpy(fd.gate_mode, st->gate_mode, sizeof(fd.gate_mode));
  fd.patch_param.num_patches = st->num_patches;
  for (i = 0; i <= XAAC_SBR_MAX_PATCHES; i++) fd.patch_param.start_subband[i] = st->patch_start_subband[i];

  rc = ixheaacd_generate_hf((FLOAT32(*)[64])(qmf_re + 128), (FLOAT32(*)[64])(qmf_im + 128),
                            ph_re ? (FLOAT32(*)[64])(ph_re + 128) : NULL, ph_im ? (FLOAT32(*)[64])(ph_im + 128) : NULL,
                            (FLOAT32(*)[64])(out_re + 128), (FLOAT32(*)[64])(out_im + 128), &fd, &hd, 0, 32, 0);
  if (rc == 0) {
    WORD32 xo[MAX_NUM_PATCHES] = {0};
    if (x_over_qmf)
      for (i = 0; i < MAX_NUM_PATCHES; i++) xo[i] = x_over_qmf[i];
    rc = ixheaacd_sbr_env_calc(&fd, (FLOAT32(*)[64])(out_re + 128), (FLOAT32(*)[64])(out_im + 128),
                               (FLOAT32(*)[64])(qmf_re + 128), (FLOAT32(*)[64])(qmf_im + 128),
                               (hd.hbe_flag && x_over_qmf) ? xo : NULL, scratch, env_out, 0, 0);
  }
  st->prev_sbr_patching_mode = fd.prev_sbr_patching_mode;

  st->env_short_flag_prev = fd.env_short_flag_prev;
  st->harm_index = fd.harm_index;
  st->phase_index = fd.phase_index;
  st->esbr_start_up = hd.esbr_start_up;
  memcpy(st->bw_array_prev, fd.bw_array_prev, sizeof(fd.bw_array_prev));
  memcpy(st->harm_flag_prev, fd.harm_flag_prev, sizeof(fd.harm_flag_prev));
  memcpy(st->e_gain, fd.e_gain, sizeof(fd.e_gain));
  memcpy(st->noise_buf, fd.noise_buf, sizeof(fd.noise_buf));
  memcpy(st->lim_table, fd.lim_table, sizeof(fd.lim_table));
  memcpy(st->gate_mode, fd.gate_mode, sizeof(fd.gate_mode));
  st->num_patches = fd.patch_param.num_patches;
  for (i = 0; i <= XAAC_SBR_MAX_PATCHES; i++) st->patch_start_subband[i] = fd.patch_param.start_subband[i];
  return rc ? -1 : 0;
}

int ref_esbr_hf_env(const xaac_sbr_header *h, const xaac_sbr_frame *f, const xaac_esbr_side *sd, xaac_esbr_state *st,
                    float *qmf_re, float *qmf_im, float *out_re, float *out_im) {
  return ref_esbr_hf_env_h(h, f, sd, st, qmf_re, qmf_im, out_re, out_im, NULL, NULL, NULL);
}

/* ---- float parametric stereo (ixheaacd_esbr_apply_ps, decoder/ixheaacd_ps_dec_flt.c:389) ---------------------------- */
/* the float members of the reference's PS ROM (ixheaacd_sbr_rom.h:193-236) by name, for tools/gen_tables_esbr_ps.py */
const void *ref_ps_flt_table(const char *name, int *count, int *is_float) {
  const ia_ps_tables_struct *t = &ixheaacd_aac_dec_ps_tables;
#define T(member, flt)                                                          \
  if (!strcmp(name, #member)) {                                                 \
    *count = (int)(sizeof(t->member) / 4);                                      \
    *is_float = flt;                                                            \
    return t->member;                                                           \
  }
  T(qmf_fract_delay_phase_factor_im, 1) T(qmf_fract_delay_phase_factor_re, 1)
  T(frac_delay_phase_fac_qmf_sub_im_20, 1) T(frac_delay_phase_fac_qmf_sub_re_20, 1)
  T(qmf_ser_fract_delay_phase_factor_im, 1) T(qmf_ser_fract_delay_phase_factor_re, 1)
  T(frac_delay_phase_fac_ser_qmf_sub_im_20, 1) T(frac_delay_phase_fac_ser_qmf_sub_re_20, 1)
  T(scale_factors_flt, 1) T(scale_factors_fine_flt, 1) T(alphas, 1) T(all_pass_link_decay_ser, 1)
  T(p8_13_20, 1) T(p2_13_20, 1) T(cos_mod_2channel, 1) T(cos_sin_mod_8channel, 1)
  T(qmf_delay_idx_tbl, 0) T(group_borders_20_tbl, 0) T(bin_group_map_20, 0) T(ipd_bins_tbl, 0)
#undef T
  if (!strcmp(name, "band_res_hyb20")) { /* WORD16[3] */
    static int32_t w[3];
    w[0] = t->band_res_hyb20[0]; w[1] = t->band_res_hyb20[1]; w[2] = t->band_res_hyb20[2];
    *count = 3;
    *is_float = 0;
    return w;
  }
  return NULL;
}

/* the real ixheaacd_esbr_apply_ps on the caller's matrices: l_* [38][64] (rows 32..37: the six look-ahead rows),
   r_* [32][64]; state in / out in the boundary format */
int ref_esbr_apply_ps(const xaac_ps_frame *pf, xaac_esbr_ps_state *st, float *l_re, float *l_im, float *r_re, float *r_im, int usb) {
  static __thread ia_ps_dec_struct ps;
  ia_ps_tables_struct *tabs = (ia_ps_tables_struct *)&ixheaacd_aac_dec_ps_tables;
  float *plr[38], *pli[38], *prr[38], *pri[38];
  static __thread float r_pad_re[38][64], r_pad_im[38][64];
  int i, j, m, k;
  memset(&ps, 0, sizeof(ps));
  ixheaacd_create_ps_esbr_dec(&ps, tabs, 64, 32, 0);
  ps.delay_sample_ser[0] = 3; ps.delay_sample_ser[1] = 4; ps.delay_sample_ser[2] = 5; /* rev_link_delay_ser, initfuncs.c:1054 */
  ps.num_env = pf->num_env;
  ps.iid_quant = pf->iid_quant;
  ps.freq_res_ipd = pf->freq_res_ipd;
  memcpy(ps.border_position, pf->border_position, sizeof(ps.border_position));
  memcpy(ps.iid_par_table, pf->iid_par_table, sizeof(ps.iid_par_table));
  memcpy(ps.icc_par_table, pf->icc_par_table, sizeof(ps.icc_par_table));
  for (i = 0; i < 3; i++)
    for (j = 0; j < 12; j++) {
      ps.hyb_qmf_buf_re_20[i][j] = st->hyb_hist_re[i][j];
      ps.hyb_qmf_buf_im_20[i][j] = st->hyb_hist_im[i][j];
    }
  memcpy(ps.qmf_delay_buf_re, st->qmf_delay_re, sizeof(st->qmf_delay_re));
  memcpy(ps.qmf_delay_buf_im, st->qmf_delay_im, sizeof(st->qmf_delay_im));
  for (i = 0; i < 2; i++)
    for (j = 0; j < 12; j++) {
      ps.sub_qmf_delay_buf_re[i][j] = st->sub_delay_re[i][j];
      ps.sub_qmf_delay_buf_im[i][j] = st->sub_delay_im[i][j];
    }
  memcpy(ps.ser_qmf_delay_buf_re, st->ser_qmf_re, sizeof(st->ser_qmf_re));
  memcpy(ps.ser_qmf_delay_buf_im, st->ser_qmf_im, sizeof(st->ser_qmf_im));
  for (m = 0; m < 3; m++)
    for (k = 0; k < 5; k++)
      for (j = 0; j < 12; j++) {
        ps.ser_sub_qmf_dealy_buf_re[m][k][j] = st->ser_sub_re[m][k][j];
        ps.ser_sub_qmf_dealy_buf_im[m][k][j] = st->ser_sub_im[m][k][j];
      }
  for (j = 0; j < 20; j++) {
    ps.h11_re_prev[j] = st->h_prev[0][j]; ps.h12_re_prev[j] = st->h_prev[1][j];
    ps.h21_re_prev[j] = st->h_prev[2][j]; ps.h22_re_prev[j] = st->h_prev[3][j];
    ps.h11_im_prev[j] = st->h_prev[4][j]; ps.h12_im_prev[j] = st->h_prev[5][j];
    ps.h21_im_prev[j] = st->h_prev[6][j]; ps.h22_im_prev[j] = st->h_prev[7][j];
    ps.peak_decay_fast_bin[j] = st->peak_decay_fast[j];
    ps.prev_nrg_bin[j] = st->prev_nrg[j];
    ps.prev_peak_diff_bin[j] = st->prev_peak_diff[j];
  }
  ps.delay_buf_idx = (WORD16)st->delay_buf_idx;
  for (m = 0; m < 3; m++) ps.delay_buf_idx_ser[m] = (WORD16)st->delay_buf_idx_ser[m];
  for (j = 0; j < 64; j++) ps.delay_qmf_delay_buf_idx[j] = st->delay_qmf_idx[j];
  for (i = 0; i < 38; i++) {
    plr[i] = l_re + 64 * i; pli[i] = l_im + 64 * i;
    prr[i] = i < 32 ? r_re + 64 * i : r_pad_re[i]; pri[i] = i < 32 ? r_im + 64 * i : r_pad_im[i];
  }
  ixheaacd_esbr_apply_ps(&ps, plr, pli, prr, pri, usb, tabs, 16);
  for (i = 0; i < 3; i++)
    for (j = 0; j < 12; j++) {
      st->hyb_hist_re[i][j] = ps.hyb_qmf_buf_re_20[i][j];
      st->hyb_hist_im[i][j] = ps.hyb_qmf_buf_im_20[i][j];
    }
  memcpy(st->qmf_delay_re, ps.qmf_delay_buf_re, sizeof(st->qmf_delay_re));
  memcpy(st->qmf_delay_im, ps.qmf_delay_buf_im, sizeof(st->qmf_delay_im));
  for (i = 0; i < 2; i++)
    for (j = 0; j < 12; j++) {
      st->sub_delay_re[i][j] = ps.sub_qmf_delay_buf_re[i][j];
      st->sub_delay_im[i][j] = ps.sub_qmf_delay_buf_im[i][j];
    }
  memcpy(st->ser_qmf_re, ps.ser_qmf_delay_buf_re, sizeof(st->ser_qmf_re));
  memcpy(st->ser_qmf_im, ps.ser_qmf_delay_buf_im, sizeof(st->ser_qmf_im));
  for (m = 0; m < 3; m++)
    for (k = 0; k < 5; k++)
      for (j = 0; j < 12; j++) {
        st->ser_sub_re[m][k][j] = ps.ser_sub_qmf_dealy_buf_re[m][k][j];
        st->ser_sub_im[m][k][j] = ps.ser_sub_qmf_dealy_buf_im[m][k][j];
      }
  for (j = 0; j < 20; j++) {
    st->h_prev[0][j] = ps.h11_re_prev[j]; st->h_prev[1][j] = ps.h12_re_prev[j];
    st->h_prev[2][j] = ps.h21_re_prev[j]; st->h_prev[3][j] = ps.h22_re_prev[j];
    st->h_prev[4][j] = ps.h11_im_prev[j]; st->h_prev[5][j] = ps.h12_im_prev[j];
    st->h_prev[6][j] = ps.h21_im_prev[j]; st->h_prev[7][j] = ps.h22_im_prev[j];
    st->peak_decay_fast[j] = ps.peak_decay_fast_bin[j];
    st->prev_nrg[j] = ps.prev_nrg_bin[j];
    st->prev_peak_diff[j] = ps.prev_peak_diff_bin[j];
  }
  st->delay_buf_idx = ps.delay_buf_idx;
  for (m = 0; m < 3; m++) st->delay_buf_idx_ser[m] = ps.delay_buf_idx_ser[m];
  for (j = 0; j < 64; j++) st->delay_qmf_idx[j] = ps.delay_qmf_delay_buf_idx[j];
  return 0;
}
