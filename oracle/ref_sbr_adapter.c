/*
 * oracle/ref_sbr_adapter.c -- TEST INFRASTRUCTURE ONLY.
 *
 * The boundary -> reference direction of the adapter: rebuilds the reference's own structs
 * (ia_sbr_dec_struct, header / frequency tables, frame data, previous-frame data, transposer
 * settings, QMF banks) from the plain-C formats of include/xaac_sbr.h, calls the REAL
 * ixheaacd_sbr_dec (decoder/ixheaacd_sbr_dec.c:662) and converts the state back.  With it the
 * tests can feed the reference arbitrary (fuzzed) side info -- inverse-filter modes, limiter
 * settings, interpolation off ... that the reference encoder never emits -- and compare with the
 * oracle / the GPU.  Linked into oracle/_ref/libref_harness.so.  No reference code is copied.
 */
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
#include "ixheaacd_audioobjtypes.h"

#include "xaac_sbr.h"

/* low-power (HE-AACv1) channel-frame through the real ixheaacd_sbr_dec */
int ref_sbr_dec_lp(const xaac_sbr_header *h, const xaac_sbr_frame *f, xaac_sbr_state *st, const int16_t *pcm_in,
                   int in_stride, int16_t *pcm_out, int out_stride) {
  static __thread ia_sbr_dec_struct d;
  static __thread ia_sbr_header_data_struct hd;
  static __thread ia_freq_band_data_struct fb;
  static __thread ia_transposer_settings_struct ts;
  static __thread ia_sbr_prev_frame_data_struct pf;
  static __thread ia_sbr_tables_struct tabs;
  static __thread WORD64 frame_mem[(sizeof(ia_sbr_frame_info_data_struct) + 1024) / 8 + 2];
  static __thread WORD32 work[64 * 48 * 2 + 1024];
  static __thread WORD32 overlap[6 * 64 * 2], lpc[2][32], copy_re[MAX_ENV_COLS][64];
  static __thread WORD16 time_data[2048 * 2];
  static __thread WORD16 ana_ring[320], syn_ring[1280], filt_me[2 * MAX_FREQ_COEFFS], filt_noise[MAX_FREQ_COEFFS];
  ia_sbr_frame_info_data_struct *fr = (ia_sbr_frame_info_data_struct *)frame_mem;
  ia_qmf_dec_tables_struct *qt = (ia_qmf_dec_tables_struct *)&ixheaacd_aac_qmf_dec_tables;
  int i, ret;
  memset(&d, 0, sizeof(d));
  memset(&hd, 0, sizeof(hd));
  memset(&fb, 0, sizeof(fb));
  memset(&ts, 0, sizeof(ts));
  memset(&pf, 0, sizeof(pf));
  memset(frame_mem, 0, sizeof(frame_mem));
  memset(&tabs, 0, sizeof(tabs));
  tabs.env_calc_tables_ptr = (ia_env_calc_tables_struct *)&ixheaacd_aac_dec_env_calc_tables;
  tabs.qmf_dec_tables_ptr = qt;
  tabs.sbr_rand_ph = (WORD32 *)ixheaacd_aac_dec_env_calc_tables.sbr_rand_ph;
  /* header + frequency tables */
  hd.num_time_slots = h->num_time_slots;
  hd.time_step = h->time_step;
  hd.channel_mode = h->channel_mode;
  hd.limiter_gains = h->limiter_gains;
  hd.interpol_freq = h->interpol_freq;
  hd.smoothing_mode = h->smoothing_mode;
  hd.pstr_freq_band_data = &fb;
  fb.num_sf_bands[0] = h->num_sf_bands[0];
  fb.num_sf_bands[1] = h->num_sf_bands[1];
  fb.num_nf_bands = h->num_nf_bands;
  fb.sub_band_start = h->sub_band_start;
  fb.sub_band_end = h->sub_band_end;
  fb.num_lf_bands = h->num_lf_bands;
  fb.num_if_bands = h->num_if_bands;
  memcpy(fb.freq_band_tbl_lim, h->freq_band_tbl_lim, sizeof(fb.freq_band_tbl_lim));
  memcpy(fb.freq_band_tbl_lo, h->freq_band_tbl_lo, sizeof(fb.freq_band_tbl_lo));
  memcpy(fb.freq_band_tbl_hi, h->freq_band_tbl_hi, sizeof(fb.freq_band_tbl_hi));
  memcpy(fb.freq_band_tbl_noise, h->freq_band_tbl_noise, sizeof(fb.freq_band_tbl_noise));
  fb.freq_band_table[0] = fb.freq_band_tbl_lo;
  fb.freq_band_table[1] = fb.freq_band_tbl_hi;
  ts.num_columns = h->num_columns;
  ts.num_patches = h->num_patches;
  ts.start_patch = h->start_patch;
  ts.stop_patch = h->stop_patch;
  memcpy(ts.bw_borders, h->bw_borders, sizeof(ts.bw_borders));
  for (i = 0; i < MAX_NUM_PATCHES; i++) memcpy(&ts.str_patch_param[i], &h->patch[i], sizeof(xaac_sbr_patch));
  /* frame side info */
  fr->str_frame_info_details.num_env = f->num_env;
  fr->str_frame_info_details.transient_env = f->transient_env;
  fr->str_frame_info_details.num_noise_env = f->num_noise_env;
  fr->str_frame_info_details.frame_class = f->frame_class;
  memcpy(fr->str_frame_info_details.border_vec, f->border_vec, sizeof(f->border_vec));
  memcpy(fr->str_frame_info_details.freq_res, f->freq_res, sizeof(f->freq_res));
  memcpy(fr->str_frame_info_details.noise_border_vec, f->noise_border_vec, sizeof(f->noise_border_vec));
  fr->amp_res = f->amp_res;
  fr->coupling_mode = f->coupling_mode;
  fr->max_qmf_subband_aac = f->max_qmf_subband_aac;
  memcpy(fr->sbr_invf_mode, f->sbr_invf_mode, sizeof(f->sbr_invf_mode));
  for (i = 0; i < MAX_FREQ_COEFFS; i++) fr->add_harmonics[i] = f->add_harmonics[i];
  memcpy(fr->int_env_sf_arr, f->int_env_sf_arr, sizeof(f->int_env_sf_arr));
  memcpy(fr->int_noise_floor, f->int_noise_floor, sizeof(f->int_noise_floor));
  /* persistent state */
  memcpy(ana_ring, st->ana_ring, sizeof(ana_ring));
  memcpy(syn_ring, st->syn_ring, sizeof(syn_ring));
  memcpy(overlap, st->overlap, sizeof(overlap));
  memcpy(lpc, st->lpc_real, sizeof(lpc));
  memcpy(filt_me, st->filt_buf_me, sizeof(filt_me));
  memcpy(filt_noise, st->filt_buf_noise_m, sizeof(filt_noise));
  d.ptr_sbr_overlap_buf = overlap;
  d.str_codec_qmf_bank.no_channels = 32;
  d.str_codec_qmf_bank.num_time_slots = 32;
  d.str_codec_qmf_bank.lsb = 0;
  d.str_codec_qmf_bank.usb = st->codec_usb;
  d.str_codec_qmf_bank.anal_filter_states = ana_ring;
  d.str_codec_qmf_bank.core_samples_buffer = ana_ring + st->ana_wr;
  d.str_codec_qmf_bank.analy_win_coeff = qt->qmf_c;
  d.str_codec_qmf_bank.filter_pos = qt->qmf_c + st->ana_phase;
  d.str_synthesis_qmf_bank.no_channels = 64;
  d.str_synthesis_qmf_bank.num_time_slots = 32;
  d.str_synthesis_qmf_bank.lsb = st->syn_lsb;
  d.str_synthesis_qmf_bank.usb = st->syn_usb;
  d.str_synthesis_qmf_bank.filter_states = syn_ring;
  d.str_synthesis_qmf_bank.p_filter = qt->qmf_c;
  d.str_synthesis_qmf_bank.filter_pos_syn = qt->qmf_c + st->syn_phase;
  d.str_synthesis_qmf_bank.ixheaacd_drc_offset = st->syn_drc_offset;
  d.str_hf_generator.pstr_settings = &ts;
  d.str_hf_generator.lpc_filt_states_real[0] = lpc[0];
  d.str_hf_generator.lpc_filt_states_real[1] = lpc[1];
  memcpy(d.str_hf_generator.bw_array_prev, st->bw_array_prev, sizeof(st->bw_array_prev));
  d.str_sbr_scale_fact.lb_scale = st->lb_scale;
  d.str_sbr_scale_fact.st_lb_scale = st->st_lb_scale;
  d.str_sbr_scale_fact.ov_lb_scale = st->ov_lb_scale;
  d.str_sbr_scale_fact.hb_scale = st->hb_scale;
  d.str_sbr_scale_fact.ov_hb_scale = st->ov_hb_scale;
  d.str_sbr_scale_fact.st_syn_scale = st->st_syn_scale;
  d.str_sbr_scale_fact.ps_scale = st->ps_scale;
  d.str_sbr_calc_env.filt_buf_me = filt_me;
  d.str_sbr_calc_env.filt_buf_noise_m = filt_noise;
  d.str_sbr_calc_env.filt_buf_noise_e = st->filt_buf_noise_e;
  d.str_sbr_calc_env.start_up = st->start_up;
  d.str_sbr_calc_env.ph_index = st->ph_index;
  d.str_sbr_calc_env.tansient_env_prev = st->tansient_env_prev;
  d.str_sbr_calc_env.harm_index = st->harm_index;
  memcpy(d.str_sbr_calc_env.harm_flags_prev, st->harm_flags_prev, sizeof(st->harm_flags_prev));
  for (i = 0; i < MAX_ENV_COLS; i++) d.p_arr_qmf_buf_real[i] = d.p_arr_qmf_buf_imag[i] = copy_re[i];
  memcpy(pf.sbr_invf_mode, st->prev_invf_mode, sizeof(pf.sbr_invf_mode));
  pf.max_qmf_subband_aac = st->prev_max_qmf_subband_aac;
  pf.coupling_mode = st->prev_coupling_mode;
  pf.end_position = st->prev_end_position;
  pf.amp_res = st->prev_amp_res;
  for (i = 0; i < 1024; i++) time_data[i] = pcm_in[(size_t)i * in_stride];
  ret = ixheaacd_sbr_dec(&d, time_data, &hd, fr, &pf, NULL, NULL, NULL, f->apply_processing, 1, work, &tabs,
                         (ixheaacd_misc_tables *)&ixheaacd_str_fft_n_transcendent_tables, 1, NULL, 0, NULL, AOT_SBR, 0,
                         NULL, 0, 0);
  for (i = 0; i < 2048; i++) pcm_out[(size_t)i * out_stride] = time_data[i];
  /* state back */
  memcpy(st->ana_ring, ana_ring, sizeof(ana_ring));
  st->ana_wr = (int16_t)(d.str_codec_qmf_bank.core_samples_buffer - ana_ring);
  st->ana_phase = (int16_t)(d.str_codec_qmf_bank.filter_pos - qt->qmf_c);
  memcpy(st->syn_ring, syn_ring, sizeof(syn_ring));
  st->syn_drc_offset = d.str_synthesis_qmf_bank.ixheaacd_drc_offset;
  st->syn_phase = (int16_t)(d.str_synthesis_qmf_bank.filter_pos_syn - qt->qmf_c);
  st->codec_usb = d.str_codec_qmf_bank.usb;
  st->syn_lsb = d.str_synthesis_qmf_bank.lsb;
  st->syn_usb = d.str_synthesis_qmf_bank.usb;
  memcpy(st->overlap, overlap, sizeof(WORD32) * 6 * 64);
  memcpy(st->lpc_real, lpc, sizeof(lpc));
  memcpy(st->bw_array_prev, d.str_hf_generator.bw_array_prev, sizeof(st->bw_array_prev));
  st->lb_scale = d.str_sbr_scale_fact.lb_scale;
  st->st_lb_scale = d.str_sbr_scale_fact.st_lb_scale;
  st->ov_lb_scale = d.str_sbr_scale_fact.ov_lb_scale;
  st->hb_scale = d.str_sbr_scale_fact.hb_scale;
  st->ov_hb_scale = d.str_sbr_scale_fact.ov_hb_scale;
  st->st_syn_scale = d.str_sbr_scale_fact.st_syn_scale;
  st->ps_scale = d.str_sbr_scale_fact.ps_scale;
  memcpy(st->prev_invf_mode, pf.sbr_invf_mode, sizeof(pf.sbr_invf_mode));
  st->prev_max_qmf_subband_aac = pf.max_qmf_subband_aac;
  st->prev_coupling_mode = pf.coupling_mode;
  st->prev_end_position = pf.end_position;
  st->prev_amp_res = pf.amp_res;
  memcpy(st->filt_buf_me, filt_me, sizeof(filt_me));
  memcpy(st->filt_buf_noise_m, filt_noise, sizeof(filt_noise));
  st->filt_buf_noise_e = d.str_sbr_calc_env.filt_buf_noise_e;
  st->start_up = d.str_sbr_calc_env.start_up;
  st->ph_index = d.str_sbr_calc_env.ph_index;
  st->tansient_env_prev = d.str_sbr_calc_env.tansient_env_prev;
  st->harm_index = d.str_sbr_calc_env.harm_index;
  memcpy(st->harm_flags_prev, d.str_sbr_calc_env.harm_flags_prev, sizeof(st->harm_flags_prev));
  return ret;
}
