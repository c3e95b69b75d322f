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

int ref_esbr_hf_env(const xaac_sbr_header *h, const xaac_sbr_frame *f, const xaac_esbr_side *sd, xaac_esbr_state *st,
                    float *qmf_re, float *qmf_im, float *out_re, float *out_im) {
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
  fd.sbr_patching_mode = 1;
  fd.prev_sbr_patching_mode = 1;
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

  rc = ixheaacd_generate_hf((FLOAT32(*)[64])(qmf_re + 128), (FLOAT32(*)[64])(qmf_im + 128), NULL, NULL,
                            (FLOAT32(*)[64])(out_re + 128), (FLOAT32(*)[64])(out_im + 128), &fd, &hd, 0, 32, 0);
  if (rc == 0)
    rc = ixheaacd_sbr_env_calc(&fd, (FLOAT32(*)[64])(out_re + 128), (FLOAT32(*)[64])(out_im + 128),
                               (FLOAT32(*)[64])(qmf_re + 128), (FLOAT32(*)[64])(qmf_im + 128), NULL, scratch, env_out, 0, 0);

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
