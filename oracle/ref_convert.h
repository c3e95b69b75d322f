/*
 * oracle/ref_convert.h -- TEST INFRASTRUCTURE ONLY.
 *
 * The field-by-field mapping between the reference decoder's own structs (read through the reference's own
 * headers) and the boundary formats of include/xaac_sbr.h, in both directions: to_*() is what a reference-side
 * integration does before it hands a channel to libxaac_amd, from_*() what it does with the state that comes back.
 * Shared by oracle/ref_capture.c (records fixtures) and oracle/ref_dropin.c (runs the reference decoder with its
 * hot path diverted to the GPU library).  No reference code is copied.
 */
#ifndef XAAC_REF_CONVERT_H
#define XAAC_REF_CONVERT_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ixheaacd_sbr_common.h"
#include "ixheaac_type_def.h"
#include "ixheaac_constants.h"
#include "ixheaac_basic_ops32.h"
#include "ixheaac_basic_ops16.h"
#include "ixheaac_basic_ops40.h"
#include "ixheaac_basic_ops.h"
#include "ixheaac_basic_op.h"
#include "ixheaacd_intrinsics.h"
#include "ixheaacd_common_rom.h"
#include "ixheaacd_sbrdecsettings.h"
#include "ixheaacd_bitbuffer.h"
#include "ixheaacd_defines.h"
#include "ixheaacd_pns.h"
#include "ixheaacd_aac_rom.h"
#include "ixheaacd_pulsedata.h"
#include "ixheaacd_drc_data_struct.h"
#include "ixheaacd_lt_predict.h"
#include "ixheaacd_cnst.h"
#include "ixheaacd_ec_defines.h"
#include "ixheaacd_ec_struct_def.h"
#include "ixheaacd_channelinfo.h"
#include "ixheaacd_drc_dec.h"
#include "ixheaacd_sbrdecoder.h"
#include "ixheaacd_definitions.h"
#include "ixheaacd_error_codes.h"
#include "ixheaacd_sbr_scale.h"
#include "ixheaacd_lpp_tran.h"
#include "ixheaacd_env_extr_part.h"
#include "ixheaacd_sbr_rom.h"
#include "ixheaacd_hybrid.h"
#include "ixheaacd_ps_dec.h"
#include "ixheaacd_env_extr.h"
#include "ixheaacd_qmf_dec.h"
#include "ixheaacd_env_calc.h"
#include "ixheaac_sbr_const.h"
#include "ixheaacd_pvc_dec.h"
#include "ixheaacd_sbr_dec.h"

#include "xaac_sbr.h"

static void to_header(const ia_sbr_header_data_struct *h, const ia_sbr_dec_struct *d, xaac_sbr_header *o) {
  const ia_freq_band_data_struct *f = h->pstr_freq_band_data;
  const ia_transposer_settings_struct *t = d->str_hf_generator.pstr_settings;
  int i;
  memset(o, 0, sizeof(*o));
  o->num_time_slots = h->num_time_slots;
  o->time_step = h->time_step;
  o->channel_mode = (int16_t)h->channel_mode;
  o->limiter_gains = h->limiter_gains;
  o->interpol_freq = h->interpol_freq;
  o->smoothing_mode = h->smoothing_mode;
  o->num_sf_bands[0] = f->num_sf_bands[0];
  o->num_sf_bands[1] = f->num_sf_bands[1];
  o->num_nf_bands = f->num_nf_bands;
  o->sub_band_start = f->sub_band_start;
  o->sub_band_end = f->sub_band_end;
  o->num_lf_bands = f->num_lf_bands;
  o->num_if_bands = f->num_if_bands;
  memcpy(o->freq_band_tbl_lim, f->freq_band_tbl_lim, sizeof(o->freq_band_tbl_lim));
  memcpy(o->freq_band_tbl_lo, f->freq_band_tbl_lo, sizeof(o->freq_band_tbl_lo));
  memcpy(o->freq_band_tbl_hi, f->freq_band_tbl_hi, sizeof(o->freq_band_tbl_hi));
  memcpy(o->freq_band_tbl_noise, f->freq_band_tbl_noise, sizeof(o->freq_band_tbl_noise));
  o->num_columns = t->num_columns;
  o->num_patches = t->num_patches;
  o->start_patch = t->start_patch;
  o->stop_patch = t->stop_patch;
  memcpy(o->bw_borders, t->bw_borders, sizeof(o->bw_borders));
  for (i = 0; i < MAX_NUM_PATCHES; i++) memcpy(&o->patch[i], &t->str_patch_param[i], sizeof(xaac_sbr_patch));
}

static void to_frame(const ia_sbr_frame_info_data_struct *f, int apply, xaac_sbr_frame *o) {
  const ia_frame_info_struct *fi = &f->str_frame_info_details;
  int i;
  memset(o, 0, sizeof(*o));
  o->num_env = fi->num_env;
  o->transient_env = fi->transient_env;
  o->num_noise_env = fi->num_noise_env;
  o->frame_class = fi->frame_class;
  memcpy(o->border_vec, fi->border_vec, sizeof(o->border_vec));
  memcpy(o->freq_res, fi->freq_res, sizeof(o->freq_res));
  memcpy(o->noise_border_vec, fi->noise_border_vec, sizeof(o->noise_border_vec));
  o->amp_res = f->amp_res;
  o->apply_processing = (int16_t)apply;
  o->coupling_mode = f->coupling_mode;
  o->max_qmf_subband_aac = f->max_qmf_subband_aac;
  memcpy(o->sbr_invf_mode, f->sbr_invf_mode, sizeof(o->sbr_invf_mode));
  for (i = 0; i < MAX_FREQ_COEFFS; i++) o->add_harmonics[i] = (uint8_t)f->add_harmonics[i];
  memcpy(o->int_env_sf_arr, f->int_env_sf_arr, sizeof(o->int_env_sf_arr));
  memcpy(o->int_noise_floor, f->int_noise_floor, sizeof(o->int_noise_floor));
}

static void to_state(const ia_sbr_dec_struct *d, const ia_sbr_prev_frame_data_struct *p, int low_pow,
                     xaac_sbr_state *o) {
  const ia_sbr_qmf_filter_bank_struct *a = &d->str_codec_qmf_bank, *s = &d->str_synthesis_qmf_bank;
  const ia_sbr_calc_env_struct *e = &d->str_sbr_calc_env;
  const ia_qmf_dec_tables_struct *t = &ixheaacd_aac_qmf_dec_tables;
  int i;
  memset(o, 0, sizeof(*o));
  memcpy(o->ana_ring, a->anal_filter_states, sizeof(o->ana_ring));
  o->ana_wr = (int16_t)(a->core_samples_buffer - a->anal_filter_states);
  o->ana_phase = (int16_t)(a->filter_pos - a->analy_win_coeff);
  memcpy(o->syn_ring, s->filter_states, sizeof(o->syn_ring));
  o->syn_drc_offset = s->ixheaacd_drc_offset;
  o->syn_phase = (int16_t)(s->filter_pos_syn - s->p_filter);
  o->codec_usb = a->usb;
  o->syn_lsb = s->lsb;
  o->syn_usb = s->usb;
  (void)t;
  memcpy(o->overlap, d->ptr_sbr_overlap_buf, sizeof(WORD32) * 64 * 6 * (low_pow ? 1 : 2));
  for (i = 0; i < 2; i++) {
    memcpy(o->lpc_real[i], d->str_hf_generator.lpc_filt_states_real[i], 32 * sizeof(WORD32));
    if (!low_pow && d->str_hf_generator.lpc_filt_states_imag[i])
      memcpy(o->lpc_imag[i], d->str_hf_generator.lpc_filt_states_imag[i], 32 * sizeof(WORD32));
  }
  memcpy(o->bw_array_prev, d->str_hf_generator.bw_array_prev, sizeof(o->bw_array_prev));
  o->lb_scale = d->str_sbr_scale_fact.lb_scale;
  o->st_lb_scale = d->str_sbr_scale_fact.st_lb_scale;
  o->ov_lb_scale = d->str_sbr_scale_fact.ov_lb_scale;
  o->hb_scale = d->str_sbr_scale_fact.hb_scale;
  o->ov_hb_scale = d->str_sbr_scale_fact.ov_hb_scale;
  o->st_syn_scale = d->str_sbr_scale_fact.st_syn_scale;
  o->ps_scale = d->str_sbr_scale_fact.ps_scale;
  memcpy(o->prev_invf_mode, p->sbr_invf_mode, sizeof(o->prev_invf_mode));
  o->prev_max_qmf_subband_aac = p->max_qmf_subband_aac;
  o->prev_coupling_mode = p->coupling_mode;
  o->prev_end_position = p->end_position;
  o->prev_amp_res = p->amp_res;
  memcpy(o->filt_buf_me, e->filt_buf_me, sizeof(o->filt_buf_me));
  memcpy(o->filt_buf_noise_m, e->filt_buf_noise_m, sizeof(o->filt_buf_noise_m));
  o->filt_buf_noise_e = e->filt_buf_noise_e;
  o->start_up = e->start_up;
  o->ph_index = e->ph_index;
  o->tansient_env_prev = e->tansient_env_prev;
  o->harm_index = e->harm_index;
  memcpy(o->harm_flags_prev, e->harm_flags_prev, sizeof(o->harm_flags_prev));
}

static void to_ps_frame(const ia_ps_dec_struct *ps, xaac_ps_frame *o) {
  memset(o, 0, sizeof(*o));
  o->iid_quant = (int16_t)ps->iid_quant;
  o->freq_res_ipd = (int16_t)ps->freq_res_ipd;
  o->num_env = ps->num_env;
  memcpy(o->border_position, ps->border_position, sizeof(o->border_position));
  memcpy(o->iid_par_table, ps->iid_par_table, sizeof(o->iid_par_table));
  memcpy(o->icc_par_table, ps->icc_par_table, sizeof(o->icc_par_table));
}

static void to_ps_state(const ia_ps_dec_struct *ps, const ia_sbr_qmf_filter_bank_struct *sr,
                        const ia_sbr_scale_fact_struct *sf_r, xaac_ps_state *o) {
  int i;
  memset(o, 0, sizeof(*o));
  memcpy(o->ser, ps->delay_buf_qmf_ser_re_im, sizeof(o->ser));
  memcpy(o->ap, ps->delay_buf_qmf_ap_re_im, sizeof(o->ap));
  memcpy(o->ld, ps->delay_buf_qmf_ld_re_im, sizeof(o->ld));
  memcpy(o->sd, ps->delay_buf_qmf_sd_re_im, 58 * sizeof(WORD16));
  memcpy(o->sub, ps->delay_buf_qmf_sub_re_im, sizeof(o->sub));
  memcpy(o->sub_ser, ps->delay_buf_qmf_sub_ser_re_im, sizeof(o->sub_ser));
  for (i = 0; i < 3; i++) {
    o->idx_ser[i] = ps->delay_buf_idx_ser[i];
    o->sample_ser[i] = ps->delay_sample_ser[i];
  }
  o->idx = ps->delay_buf_idx;
  o->idx_long = ps->delay_buf_idx_long;
  memcpy(o->peak_decay_diff, ps->peak_decay_diff, sizeof(o->peak_decay_diff));
  memcpy(o->energy_prev, ps->energy_prev, sizeof(o->energy_prev));
  memcpy(o->peak_decay_diff_prev, ps->peak_decay_diff_prev, sizeof(o->peak_decay_diff_prev));
  for (i = 0; i < 3; i++) {
    memcpy(o->hyb_buf[i][0], ps->str_hybrid.ptr_qmf_buf_re[i], 12 * sizeof(WORD32));
    memcpy(o->hyb_buf[i][1], ps->str_hybrid.ptr_qmf_buf_im[i], 12 * sizeof(WORD32));
  }
  memcpy(o->h11_h12_vec, ps->h11_h12_vec, sizeof(o->h11_h12_vec));
  memcpy(o->h21_h22_vec, ps->h21_h22_vec, sizeof(o->h21_h22_vec));
  memcpy(o->H11_H12, ps->H11_H12, sizeof(o->H11_H12));
  memcpy(o->H21_H22, ps->H21_H22, sizeof(o->H21_H22));
  memcpy(o->delta_h11_h12, ps->delta_h11_h12, sizeof(o->delta_h11_h12));
  memcpy(o->delta_h21_h22, ps->delta_h21_h22, sizeof(o->delta_h21_h22));
  o->delay_buffer_scale = ps->delay_buffer_scale;
  o->usb = ps->usb;
  memcpy(o->syn_ring_r, sr->filter_states, sizeof(o->syn_ring_r));
  o->syn_drc_offset_r = sr->ixheaacd_drc_offset;
  o->syn_phase_r = (int16_t)(sr->filter_pos_syn - sr->p_filter);
  o->syn_lsb_r = sr->lsb;
  o->syn_usb_r = sr->usb;
  o->st_syn_scale_r = sf_r->st_syn_scale;
  o->lb_scale_r = sf_r->lb_scale;
  o->ov_lb_scale_r = sf_r->ov_lb_scale;
  o->hb_scale_r = sf_r->hb_scale;
}


/* ---- the way back: boundary state -> the live reference structs -------------------------------- */
static void from_state(const xaac_sbr_state *o, int low_pow, ia_sbr_dec_struct *d, ia_sbr_prev_frame_data_struct *p) {
  ia_sbr_qmf_filter_bank_struct *a = &d->str_codec_qmf_bank, *s = &d->str_synthesis_qmf_bank;
  ia_sbr_calc_env_struct *e = &d->str_sbr_calc_env;
  int i;
  memcpy(a->anal_filter_states, o->ana_ring, sizeof(o->ana_ring));
  a->core_samples_buffer = a->anal_filter_states + o->ana_wr;
  a->filter_pos = (WORD16 *)a->analy_win_coeff + o->ana_phase;
  memcpy(s->filter_states, o->syn_ring, sizeof(o->syn_ring));
  s->ixheaacd_drc_offset = o->syn_drc_offset;
  s->filter_pos_syn = (WORD16 *)s->p_filter + o->syn_phase;
  a->usb = o->codec_usb;
  s->lsb = o->syn_lsb;
  s->usb = o->syn_usb;
  memcpy(d->ptr_sbr_overlap_buf, o->overlap, sizeof(WORD32) * 64 * 6 * (low_pow ? 1 : 2));
  for (i = 0; i < 2; i++) {
    memcpy(d->str_hf_generator.lpc_filt_states_real[i], o->lpc_real[i], 32 * sizeof(WORD32));
    if (!low_pow && d->str_hf_generator.lpc_filt_states_imag[i])
      memcpy(d->str_hf_generator.lpc_filt_states_imag[i], o->lpc_imag[i], 32 * sizeof(WORD32));
  }
  memcpy(d->str_hf_generator.bw_array_prev, o->bw_array_prev, sizeof(o->bw_array_prev));
  d->str_sbr_scale_fact.lb_scale = o->lb_scale;
  d->str_sbr_scale_fact.st_lb_scale = o->st_lb_scale;
  d->str_sbr_scale_fact.ov_lb_scale = o->ov_lb_scale;
  d->str_sbr_scale_fact.hb_scale = o->hb_scale;
  d->str_sbr_scale_fact.ov_hb_scale = o->ov_hb_scale;
  d->str_sbr_scale_fact.st_syn_scale = o->st_syn_scale;
  d->str_sbr_scale_fact.ps_scale = o->ps_scale;
  memcpy(p->sbr_invf_mode, o->prev_invf_mode, sizeof(o->prev_invf_mode));
  p->max_qmf_subband_aac = o->prev_max_qmf_subband_aac;
  p->coupling_mode = o->prev_coupling_mode;
  p->end_position = o->prev_end_position;
  p->amp_res = o->prev_amp_res;
  memcpy(e->filt_buf_me, o->filt_buf_me, sizeof(o->filt_buf_me));
  memcpy(e->filt_buf_noise_m, o->filt_buf_noise_m, sizeof(o->filt_buf_noise_m));
  e->filt_buf_noise_e = o->filt_buf_noise_e;
  e->start_up = o->start_up;
  e->ph_index = o->ph_index;
  e->tansient_env_prev = o->tansient_env_prev;
  e->harm_index = o->harm_index;
  memcpy(e->harm_flags_prev, o->harm_flags_prev, sizeof(o->harm_flags_prev));
}

/* ---- AAC-ELD channels (low-delay SBR): xaac_sbr_eld_state <-> the reference's structs.  The members ixheaacd_sbr_dec's core
   works on are the ones to_state / from_state carry; the LD banks keep four rotating pointers each (generic:609-741, qmf_dec.c:
   862-1135), stored as offsets.  `t`: the QMF tables (the banks are re-based onto qmf_c_eld3 / qmf_c_eld first, as the two
   functions do on entry). */
#include <stddef.h>
#include "xaac_amd.h"
static void to_eld_state(ia_sbr_dec_struct *d, const ia_sbr_prev_frame_data_struct *p, ia_qmf_dec_tables_struct *t, xaac_sbr_eld_state *o) {
  static xaac_sbr_state tmp;
  ia_sbr_qmf_filter_bank_struct *a = &d->str_codec_qmf_bank, *s = &d->str_synthesis_qmf_bank;
  to_state(d, p, 0, &tmp);
  memset(o, 0, sizeof(*o));
  o->codec_usb = tmp.codec_usb, o->syn_lsb = tmp.syn_lsb, o->syn_usb = tmp.syn_usb;
  memcpy(&o->lpc_real, &tmp.lpc_real, sizeof(xaac_sbr_state) - offsetof(xaac_sbr_state, lpc_real));
  a->filter_pos += t->qmf_c_eld3 - a->analy_win_coeff;
  a->analy_win_coeff = t->qmf_c_eld3;
  memcpy(o->ana.ring, a->anal_filter_states, sizeof(o->ana.ring));
  o->ana.wr = (int16_t)(a->core_samples_buffer - a->anal_filter_states);
  o->ana.f1 = (int16_t)(a->filter_pos - t->qmf_c_eld3);
  o->ana.f2 = (int16_t)(a->filter_2 - t->qmf_c_eld3);
  o->ana.fp = (int16_t)(a->fp1_anal - a->anal_filter_states);
  s->filter_pos_syn += t->qmf_c_eld - s->p_filter;
  s->p_filter = t->qmf_c_eld;
  memcpy(o->syn.ring, s->filter_states, sizeof(o->syn.ring));
  o->syn.drc_offset = (int16_t)s->ixheaacd_drc_offset;
  o->syn.phase = (int16_t)(s->filter_pos_syn - t->qmf_c_eld);
  o->syn.fp = (int16_t)(s->fp1_syn - s->filter_states);
  o->syn.sixty4 = (int16_t)s->sixty4;
}
static void from_eld_state(const xaac_sbr_eld_state *o, ia_sbr_dec_struct *d, ia_sbr_prev_frame_data_struct *p, ia_qmf_dec_tables_struct *t) {
  static xaac_sbr_state tmp;
  ia_sbr_qmf_filter_bank_struct *a = &d->str_codec_qmf_bank, *s = &d->str_synthesis_qmf_bank;
  to_state(d, p, 0, &tmp); /* (the AAC banks' words and the overlap buffer go back as they came) */
  tmp.codec_usb = o->codec_usb, tmp.syn_lsb = o->syn_lsb, tmp.syn_usb = o->syn_usb;
  memcpy(&tmp.lpc_real, &o->lpc_real, sizeof(xaac_sbr_state) - offsetof(xaac_sbr_state, lpc_real));
  from_state(&tmp, 0, d, p);
  memcpy(a->anal_filter_states, o->ana.ring, sizeof(o->ana.ring));
  a->core_samples_buffer = a->anal_filter_states + o->ana.wr;
  a->filter_pos = t->qmf_c_eld3 + o->ana.f1;
  a->filter_2 = t->qmf_c_eld3 + o->ana.f2;
  a->fp1_anal = a->anal_filter_states + o->ana.fp;
  a->fp2_anal = a->anal_filter_states + (32 - o->ana.fp);
  memcpy(s->filter_states, o->syn.ring, sizeof(o->syn.ring));
  s->ixheaacd_drc_offset = o->syn.drc_offset;
  s->filter_pos_syn = t->qmf_c_eld + o->syn.phase;
  s->fp1_syn = s->filter_states + o->syn.fp;
  s->sixty4 = o->syn.sixty4;
  s->fp2_syn = s->fp1_syn + s->sixty4;
}

static void from_ps_state(const xaac_ps_state *o, ia_ps_dec_struct *ps, ia_sbr_qmf_filter_bank_struct *sr,
                          ia_sbr_scale_fact_struct *sf_r) {
  int i;
  memcpy(ps->delay_buf_qmf_ser_re_im, o->ser, sizeof(o->ser));
  memcpy(ps->delay_buf_qmf_ap_re_im, o->ap, sizeof(o->ap));
  memcpy(ps->delay_buf_qmf_ld_re_im, o->ld, sizeof(o->ld));
  memcpy(ps->delay_buf_qmf_sd_re_im, o->sd, 58 * sizeof(WORD16));
  memcpy(ps->delay_buf_qmf_sub_re_im, o->sub, sizeof(o->sub));
  memcpy(ps->delay_buf_qmf_sub_ser_re_im, o->sub_ser, sizeof(o->sub_ser));
  for (i = 0; i < 3; i++) ps->delay_buf_idx_ser[i] = o->idx_ser[i];
  ps->delay_buf_idx = o->idx;
  ps->delay_buf_idx_long = o->idx_long;
  memcpy(ps->peak_decay_diff, o->peak_decay_diff, sizeof(o->peak_decay_diff));
  memcpy(ps->energy_prev, o->energy_prev, sizeof(o->energy_prev));
  memcpy(ps->peak_decay_diff_prev, o->peak_decay_diff_prev, sizeof(o->peak_decay_diff_prev));
  for (i = 0; i < 3; i++) {
    memcpy(ps->str_hybrid.ptr_qmf_buf_re[i], o->hyb_buf[i][0], 12 * sizeof(WORD32));
    memcpy(ps->str_hybrid.ptr_qmf_buf_im[i], o->hyb_buf[i][1], 12 * sizeof(WORD32));
  }
  memcpy(ps->h11_h12_vec, o->h11_h12_vec, sizeof(o->h11_h12_vec));
  memcpy(ps->h21_h22_vec, o->h21_h22_vec, sizeof(o->h21_h22_vec));
  memcpy(ps->H11_H12, o->H11_H12, sizeof(o->H11_H12));
  memcpy(ps->H21_H22, o->H21_H22, sizeof(o->H21_H22));
  memcpy(ps->delta_h11_h12, o->delta_h11_h12, sizeof(o->delta_h11_h12));
  memcpy(ps->delta_h21_h22, o->delta_h21_h22, sizeof(o->delta_h21_h22));
  ps->delay_buffer_scale = o->delay_buffer_scale;
  ps->usb = o->usb;
  memcpy(sr->filter_states, o->syn_ring_r, sizeof(o->syn_ring_r));
  sr->ixheaacd_drc_offset = o->syn_drc_offset_r;
  sr->filter_pos_syn = (WORD16 *)sr->p_filter + o->syn_phase_r;
  sf_r->lb_scale = o->lb_scale_r;
  sf_r->ov_lb_scale = o->ov_lb_scale_r;
  sf_r->hb_scale = o->hb_scale_r;
}

/* ---- Path A (eSBR): include/xaac_esbr.h <-> the members ixheaacd_sbr_dec's enh_sbr branch works on ----------------- */
#include "xaac_esbr.h"

static void to_esbr_side(const ia_sbr_header_data_struct *h, const ia_sbr_frame_info_data_struct *f, xaac_esbr_side *o) {
  const ia_freq_band_data_struct *fb = h->pstr_freq_band_data;
  int i;
  memset(o, 0, sizeof(*o));
  o->out_sampling_freq = h->out_sampling_freq;
  o->limiter_bands = h->limiter_bands;
  o->num_mf_bands = fb->num_mf_bands;
  memcpy(o->f_master_tbl, fb->f_master_tbl, sizeof(o->f_master_tbl));
  o->qmf_sb_prev = fb->qmf_sb_prev;
  o->reset_flag = (int16_t)f->reset_flag;
  for (i = 0; i < XAAC_SBR_MAX_NOISE_VALUES; i++) o->sbr_invf_mode_prev[i] = f->sbr_invf_mode_prev[i];
  for (i = 0; i < XAAC_SBR_MAX_ENVELOPES; i++) o->inter_temp_shape_mode[i] = f->inter_temp_shape_mode[i];
  memcpy(o->flt_env_sf_arr, f->flt_env_sf_arr, sizeof(o->flt_env_sf_arr));
  memcpy(o->flt_noise_floor, f->flt_noise_floor, sizeof(o->flt_noise_floor));
  o->harmonic_sbr = (int16_t)((f->sbr_patching_mode == 0 ? XAAC_ESBR_HARMONIC : 0) | (h->pre_proc_flag ? XAAC_ESBR_PRE_FLATTEN : 0) |
                              (h->usac_flag ? XAAC_ESBR_USAC : 0) | (f->over_sampling_flag ? XAAC_ESBR_OVERSAMPLING : 0) |
                              (h->usac_flag && !h->hbe_flag ? XAAC_ESBR_NO_X_DELAY : 0) | /* sbr_dec.c:819-826: codec_x_delay */
                              (f->sbr_mode != ORIG_SBR && f->sbr_mode != PVC_SBR ? XAAC_ESBR_SKIP_ADJUST : 0));
  o->pitch_in_bins = f->pitch_in_bins;
}

/* ---- PVC (USAC channels): xaac_esbr_pvc_side / xaac_esbr_pvc_state <-> the reference's frame data, header and ia_pvc_data_struct */
static void to_esbr_pvc_side(const ia_sbr_header_data_struct *h, const ia_sbr_frame_info_data_struct *f, const ia_pvc_data_struct *pvc,
                             FLAG low_pow, xaac_esbr_pvc_side *o) {
  int i;
  memset(o, 0, sizeof(*o));
  o->sbr_mode = (int16_t)f->sbr_mode;
  o->sine_position = (int16_t)f->sine_position;
  o->sin_start_for_cur_top = (int16_t)f->sin_start_for_cur_top;
  o->sin_len_for_cur_top = (int16_t)f->sin_len_for_cur_top;
  for (i = 0; i <= XAAC_SBR_MAX_ENVELOPES; i++) o->border_vec[i] = f->str_pvc_frame_info.border_vec[i];
  for (i = 0; i < XAAC_SBR_MAX_ENVELOPES; i++) o->freq_res[i] = f->str_pvc_frame_info.freq_res[i];
  if (pvc) {
    o->pvc.pvc_mode = pvc->pvc_mode;
    o->pvc.ns_mode = pvc->ns_mode;
    o->pvc.pvc_rate = (uint8_t)h->upsamp_fac; /* sbr_dec.c:931 */
    o->pvc.low_power = (uint8_t)(low_pow != 0);
    o->pvc.first_bnd_idx = h->pstr_freq_band_data->sub_band_start;
    o->pvc.first_pvc_timeslot = f->str_pvc_frame_info.border_vec[0];
    for (i = 0; i < XAAC_PVC_SLOTS; i++) o->pvc.pvc_id[i] = pvc->pvc_id[i];
  }
}
static void to_esbr_pvc_state(const ia_sbr_header_data_struct *h, const ia_sbr_frame_info_data_struct *f, const ia_pvc_data_struct *pvc,
                              xaac_esbr_pvc_state *o) {
  memset(o, 0, sizeof(*o));
  if (pvc) {
    memcpy(o->pvc.esg, pvc->esg, sizeof(o->pvc.esg));
    o->pvc.prev_first_bnd_idx = pvc->prev_first_bnd_idx;
    o->pvc.prev_pvc_id = pvc->prev_pvc_id;
    o->pvc.prev_pvc_flg = pvc->prev_pvc_flg;
    o->pvc.prev_pvc_rate = pvc->prev_pvc_rate;
  }
  memcpy(o->qmapped, f->qmapped_pvc, sizeof(o->qmapped));
  memcpy(o->prev_noise_level, f->prev_noise_level, sizeof(o->prev_noise_level));
  memcpy(o->harm_flag_varlen_prev, f->harm_flag_varlen_prev, 64);
  memcpy(o->harm_flag_varlen, f->harm_flag_varlen, 64);
  o->prev_freq_res[0] = f->str_frame_info_prev.freq_res[0];
  o->prev_freq_res[1] = f->str_frame_info_prev.freq_res[1];
  o->var_len_id_prev = (int16_t)f->var_len_id_prev;
  o->prev_sbr_mode = (int16_t)f->prev_sbr_mode;
  o->esbr_start_up_pvc = h->esbr_start_up_pvc;
}
/* back into the reference's structs; `processed`: the call ran its tools (apply_processing and no refusal) */
static void from_esbr_pvc_state(const xaac_esbr_pvc_state *o, int processed, ia_sbr_header_data_struct *h, ia_sbr_frame_info_data_struct *f,
                                ia_pvc_data_struct *pvc) {
  if (pvc && processed) {
    memcpy(pvc->esg, o->pvc.esg, sizeof(o->pvc.esg));
    pvc->prev_first_bnd_idx = o->pvc.prev_first_bnd_idx;
    pvc->prev_pvc_id = o->pvc.prev_pvc_id;
    pvc->prev_pvc_flg = o->pvc.prev_pvc_flg;
    pvc->prev_pvc_rate = o->pvc.prev_pvc_rate;
    pvc->pvc_rate = (UWORD8)h->upsamp_fac;
  }
  memcpy(f->qmapped_pvc, o->qmapped, sizeof(o->qmapped));
  memcpy(f->prev_noise_level, o->prev_noise_level, sizeof(o->prev_noise_level));
  memcpy(f->harm_flag_varlen_prev, o->harm_flag_varlen_prev, 64);
  memcpy(f->harm_flag_varlen, o->harm_flag_varlen, 64);
  if (processed) { /* (str_frame_info_prev is a copy of the frame's own grid: esbr_envcal.c:873) */
    memcpy(&f->str_frame_info_prev, &f->str_frame_info_details, sizeof(ia_frame_info_struct));
  }
  f->var_len_id_prev = o->var_len_id_prev;
  f->prev_sbr_mode = o->prev_sbr_mode;
  h->esbr_start_up_pvc = o->esbr_start_up_pvc;
}

/* the state as the call finds it: this library's rows 0.. are the rows the reference is about to move down from row 32 */
static void to_esbr_state(const ia_sbr_dec_struct *d, const ia_sbr_header_data_struct *h,
                          const ia_sbr_frame_info_data_struct *f, xaac_esbr_state *o) {
  const ia_sbr_qmf_filter_bank_struct *a = &d->str_codec_qmf_bank, *s = &d->str_synthesis_qmf_bank;
  const int nts = a->num_time_slots == 64 ? 64 : 32; /* 4:1 SBR: the next frame's history starts at row 64 (sbr_dec.c:835-857) */
  int i;
  memset(o, 0, sizeof(*o));
  memcpy(o->ana.ring, a->anal_filter_states_32, sizeof(o->ana.ring));
  o->ana.pos = (int32_t)(a->state_new_samples_pos_low_32 - a->anal_filter_states_32);
  o->ana.win_off = (int32_t)(a->filter_pos_32 - a->analy_win_coeff_32);
  memcpy(o->syn.ring, s->filter_states_32, sizeof(o->syn.ring));
  o->syn.drc_offset = s->ixheaacd_drc_offset;
  o->syn.filt_off = (int32_t)(s->filter_pos_syn_32 - s->p_filter_32);
  memcpy(o->qmf_re, d->qmf_buf_real[nts], sizeof(o->qmf_re));
  memcpy(o->qmf_im, d->qmf_buf_imag[nts], sizeof(o->qmf_im));
  memcpy(o->out_re, d->sbr_qmf_out_real[nts], sizeof(o->out_re));
  memcpy(o->out_im, d->sbr_qmf_out_imag[nts], sizeof(o->out_im));
  memcpy(o->bw_array_prev, f->bw_array_prev, sizeof(o->bw_array_prev));
  memcpy(o->e_gain, f->e_gain, sizeof(o->e_gain));
  memcpy(o->noise_buf, f->noise_buf, sizeof(o->noise_buf));
  memcpy(o->lim_table, f->lim_table, sizeof(o->lim_table));
  memcpy(o->gate_mode, f->gate_mode, sizeof(o->gate_mode));
  o->harm_index = f->harm_index;
  o->phase_index = f->phase_index;
  o->esbr_start_up = h->esbr_start_up;
  o->env_short_flag_prev = f->env_short_flag_prev;
  o->num_patches = f->patch_param.num_patches;
  for (i = 0; i <= XAAC_SBR_MAX_PATCHES; i++) o->patch_start_subband[i] = f->patch_param.start_subband[i];
  memcpy(o->harm_flag_prev, f->harm_flag_prev, sizeof(o->harm_flag_prev));
  o->prev_sbr_patching_mode = f->prev_sbr_patching_mode;
  memcpy(o->ph_re, d->ph_vocod_qmf_real[32], sizeof(o->ph_re));
  memcpy(o->ph_im, d->ph_vocod_qmf_imag[32], sizeof(o->ph_im));
  if (nts == 64) { /* xaac_esbr.h, XAAC_ESBR_OUT_HIST_ROWS_4_1: rows 8..13 of sbr_qmf_out's 14-row history ride in the ph rows */
    memset(o->ph_re, 0, sizeof(o->ph_re));
    memset(o->ph_im, 0, sizeof(o->ph_im));
    memcpy(o->ph_re, d->sbr_qmf_out_real[nts + 8], sizeof(float) * 6 * 64);
    memcpy(o->ph_im, d->sbr_qmf_out_imag[nts + 8], sizeof(float) * 6 * 64);
  }
}

static void from_esbr_state(const xaac_esbr_state *o, ia_sbr_dec_struct *d, ia_sbr_header_data_struct *h,
                            ia_sbr_frame_info_data_struct *f) {
  ia_sbr_qmf_filter_bank_struct *a = &d->str_codec_qmf_bank, *s = &d->str_synthesis_qmf_bank;
  const int nts = a->num_time_slots == 64 ? 64 : 32;
  int i;
  memcpy(a->anal_filter_states_32, o->ana.ring, sizeof(o->ana.ring));
  a->state_new_samples_pos_low_32 = a->anal_filter_states_32 + o->ana.pos;
  a->filter_pos_32 = a->analy_win_coeff_32 + o->ana.win_off;
  memcpy(s->filter_states_32, o->syn.ring, sizeof(o->syn.ring));
  s->ixheaacd_drc_offset = o->syn.drc_offset;
  s->filter_pos_syn_32 = (WORD32 *)s->p_filter_32 + o->syn.filt_off;
  memcpy(d->qmf_buf_real[nts], o->qmf_re, sizeof(o->qmf_re));
  memcpy(d->qmf_buf_imag[nts], o->qmf_im, sizeof(o->qmf_im));
  memcpy(d->sbr_qmf_out_real[nts], o->out_re, sizeof(o->out_re));
  memcpy(d->sbr_qmf_out_imag[nts], o->out_im, sizeof(o->out_im));
  memcpy(f->bw_array_prev, o->bw_array_prev, sizeof(o->bw_array_prev));
  memcpy(f->e_gain, o->e_gain, sizeof(o->e_gain));
  memcpy(f->noise_buf, o->noise_buf, sizeof(o->noise_buf));
  memcpy(f->lim_table, o->lim_table, sizeof(o->lim_table));
  memcpy(f->gate_mode, o->gate_mode, sizeof(o->gate_mode));
  f->harm_index = o->harm_index;
  f->phase_index = o->phase_index;
  h->esbr_start_up = o->esbr_start_up;
  f->env_short_flag_prev = o->env_short_flag_prev;
  f->patch_param.num_patches = o->num_patches;
  for (i = 0; i <= XAAC_SBR_MAX_PATCHES; i++) f->patch_param.start_subband[i] = o->patch_start_subband[i];
  memcpy(f->harm_flag_prev, o->harm_flag_prev, sizeof(o->harm_flag_prev));
  f->prev_sbr_patching_mode = o->prev_sbr_patching_mode;
  if (nts == 64) {
    memcpy(d->sbr_qmf_out_real[nts + 8], o->ph_re, sizeof(float) * 6 * 64);
    memcpy(d->sbr_qmf_out_imag[nts + 8], o->ph_im, sizeof(float) * 6 * 64);
  } else {
    memcpy(d->ph_vocod_qmf_real[32], o->ph_re, sizeof(o->ph_re));
    memcpy(d->ph_vocod_qmf_imag[32], o->ph_im, sizeof(o->ph_im));
  }
}

/* the QMF harmonic transposer (ia_esbr_hbe_txposer_struct, ixheaacd_sbr_dec.h:30-100) <-> xaac_hbe_state */
static void to_hbe_state(const ia_esbr_hbe_txposer_struct *t, xaac_hbe_state *o) {
  int i;
  memset(o, 0, sizeof(*o));
  memcpy(o->input_buf, t->ptr_input_buf, sizeof(o->input_buf));
  memcpy(o->synth_buf, t->synth_buf, sizeof(o->synth_buf));
  memcpy(o->analy_buf, t->analy_buf, sizeof(o->analy_buf));
  for (i = 0; i < XAAC_HBE_NO_BINS; i++) memcpy(o->qmf_in_buf[i], t->qmf_in_buf[i], sizeof(o->qmf_in_buf[i]));
  for (i = 0; i < 2 * XAAC_HBE_NO_BINS; i++) memcpy(o->qmf_out_buf[i], t->qmf_out_buf[i], sizeof(o->qmf_out_buf[i]));
  o->synth_size = t->synth_size;
  o->k_start = t->k_start;
  o->start_band = t->start_band;
  o->end_band = t->end_band;
  for (i = 0; i < 6; i++) o->x_over_qmf[i] = t->x_over_qmf[i];
  o->max_stretch = t->max_stretch;
  o->fft_ready = t->ixheaacd_cmplx_anal_fft != NULL;
}

/* hd: the header whose tables the reference's own re-initialisation inside ixheaacd_qmf_hbe_apply would have used
   (hbe_trans.c:240-248) -- run here when this library's call made the transposer's FFTs "ready", so that the
   reference's struct gets its function and table pointers */
static void from_hbe_state(const xaac_hbe_state *o, ia_esbr_hbe_txposer_struct *t, ia_sbr_header_data_struct *hd) {
  int i;
  if (o->fft_ready && t->ixheaacd_cmplx_anal_fft == NULL)
    ixheaacd_qmf_hbe_data_reinit(t, hd->pstr_freq_band_data->freq_band_table, hd->pstr_freq_band_data->num_sf_bands,
                                 hd->is_usf_4);
  memcpy(t->ptr_input_buf, o->input_buf, sizeof(o->input_buf));
  memcpy(t->synth_buf, o->synth_buf, sizeof(o->synth_buf));
  memcpy(t->analy_buf, o->analy_buf, sizeof(o->analy_buf));
  for (i = 0; i < XAAC_HBE_NO_BINS; i++) memcpy(t->qmf_in_buf[i], o->qmf_in_buf[i], sizeof(o->qmf_in_buf[i]));
  for (i = 0; i < 2 * XAAC_HBE_NO_BINS; i++) memcpy(t->qmf_out_buf[i], o->qmf_out_buf[i], sizeof(o->qmf_out_buf[i]));
}

static void to_esbr_ps_state(const ia_ps_dec_struct *ps, const ia_sbr_qmf_filter_bank_struct *sr, xaac_esbr_ps_state *o) {
  int i, j, m, k;
  memset(o, 0, sizeof(*o));
  for (i = 0; i < 3; i++)
    for (j = 0; j < 12; j++) {
      o->hyb_hist_re[i][j] = ps->hyb_qmf_buf_re_20[i][j];
      o->hyb_hist_im[i][j] = ps->hyb_qmf_buf_im_20[i][j];
    }
  memcpy(o->qmf_delay_re, ps->qmf_delay_buf_re, sizeof(o->qmf_delay_re));
  memcpy(o->qmf_delay_im, ps->qmf_delay_buf_im, sizeof(o->qmf_delay_im));
  for (i = 0; i < 2; i++)
    for (j = 0; j < 12; j++) {
      o->sub_delay_re[i][j] = ps->sub_qmf_delay_buf_re[i][j];
      o->sub_delay_im[i][j] = ps->sub_qmf_delay_buf_im[i][j];
    }
  memcpy(o->ser_qmf_re, ps->ser_qmf_delay_buf_re, sizeof(o->ser_qmf_re));
  memcpy(o->ser_qmf_im, ps->ser_qmf_delay_buf_im, sizeof(o->ser_qmf_im));
  for (m = 0; m < 3; m++)
    for (k = 0; k < 5; k++)
      for (j = 0; j < 12; j++) {
        o->ser_sub_re[m][k][j] = ps->ser_sub_qmf_dealy_buf_re[m][k][j];
        o->ser_sub_im[m][k][j] = ps->ser_sub_qmf_dealy_buf_im[m][k][j];
      }
  for (j = 0; j < 20; j++) {
    o->h_prev[0][j] = ps->h11_re_prev[j]; o->h_prev[1][j] = ps->h12_re_prev[j];
    o->h_prev[2][j] = ps->h21_re_prev[j]; o->h_prev[3][j] = ps->h22_re_prev[j];
    o->h_prev[4][j] = ps->h11_im_prev[j]; o->h_prev[5][j] = ps->h12_im_prev[j];
    o->h_prev[6][j] = ps->h21_im_prev[j]; o->h_prev[7][j] = ps->h22_im_prev[j];
    o->peak_decay_fast[j] = ps->peak_decay_fast_bin[j];
    o->prev_nrg[j] = ps->prev_nrg_bin[j];
    o->prev_peak_diff[j] = ps->prev_peak_diff_bin[j];
  }
  o->delay_buf_idx = ps->delay_buf_idx;
  for (m = 0; m < 3; m++) o->delay_buf_idx_ser[m] = ps->delay_buf_idx_ser[m];
  for (j = 0; j < 64; j++) o->delay_qmf_idx[j] = ps->delay_qmf_delay_buf_idx[j];
  memcpy(o->syn_r.ring, sr->filter_states_32, sizeof(o->syn_r.ring));
  o->syn_r.drc_offset = sr->ixheaacd_drc_offset;
  o->syn_r.filt_off = (int32_t)(sr->filter_pos_syn_32 - sr->p_filter_32);
}

static void from_esbr_ps_state(const xaac_esbr_ps_state *o, ia_ps_dec_struct *ps, ia_sbr_qmf_filter_bank_struct *sr) {
  int i, j, m, k;
  for (i = 0; i < 3; i++)
    for (j = 0; j < 12; j++) {
      ps->hyb_qmf_buf_re_20[i][j] = o->hyb_hist_re[i][j];
      ps->hyb_qmf_buf_im_20[i][j] = o->hyb_hist_im[i][j];
    }
  memcpy(ps->qmf_delay_buf_re, o->qmf_delay_re, sizeof(o->qmf_delay_re));
  memcpy(ps->qmf_delay_buf_im, o->qmf_delay_im, sizeof(o->qmf_delay_im));
  for (i = 0; i < 2; i++)
    for (j = 0; j < 12; j++) {
      ps->sub_qmf_delay_buf_re[i][j] = o->sub_delay_re[i][j];
      ps->sub_qmf_delay_buf_im[i][j] = o->sub_delay_im[i][j];
    }
  memcpy(ps->ser_qmf_delay_buf_re, o->ser_qmf_re, sizeof(o->ser_qmf_re));
  memcpy(ps->ser_qmf_delay_buf_im, o->ser_qmf_im, sizeof(o->ser_qmf_im));
  for (m = 0; m < 3; m++)
    for (k = 0; k < 5; k++)
      for (j = 0; j < 12; j++) {
        ps->ser_sub_qmf_dealy_buf_re[m][k][j] = o->ser_sub_re[m][k][j];
        ps->ser_sub_qmf_dealy_buf_im[m][k][j] = o->ser_sub_im[m][k][j];
      }
  for (j = 0; j < 20; j++) {
    ps->h11_re_prev[j] = o->h_prev[0][j]; ps->h12_re_prev[j] = o->h_prev[1][j];
    ps->h21_re_prev[j] = o->h_prev[2][j]; ps->h22_re_prev[j] = o->h_prev[3][j];
    ps->h11_im_prev[j] = o->h_prev[4][j]; ps->h12_im_prev[j] = o->h_prev[5][j];
    ps->h21_im_prev[j] = o->h_prev[6][j]; ps->h22_im_prev[j] = o->h_prev[7][j];
    ps->peak_decay_fast_bin[j] = o->peak_decay_fast[j];
    ps->prev_nrg_bin[j] = o->prev_nrg[j];
    ps->prev_peak_diff_bin[j] = o->prev_peak_diff[j];
  }
  ps->delay_buf_idx = (WORD16)o->delay_buf_idx;
  for (m = 0; m < 3; m++) ps->delay_buf_idx_ser[m] = (WORD16)o->delay_buf_idx_ser[m];
  for (j = 0; j < 64; j++) ps->delay_qmf_delay_buf_idx[j] = o->delay_qmf_idx[j];
  memcpy(sr->filter_states_32, o->syn_r.ring, sizeof(o->syn_r.ring));
  sr->ixheaacd_drc_offset = o->syn_r.drc_offset;
  sr->filter_pos_syn_32 = (WORD32 *)sr->p_filter_32 + o->syn_r.filt_off;
}

#endif
