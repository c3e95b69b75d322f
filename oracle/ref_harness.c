/*
 * oracle/ref_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin C entry points around the *real* reference functions so that Python
 * (ctypes) can drive them without mirroring the reference's structs.  Compiled
 * against the reference headers where they lie under /root/reference and linked
 * with the reference objects into oracle/_ref/libref_harness.so by
 * oracle/Makefile.ref.  Contains no reference code: it only fills the reference's
 * own structs (as decoder/ixheaacd_aacdecoder.c:192-209 does for the window
 * pointers) and calls the reference's external symbols.
 */
#include <stdio.h>
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
#include "ixheaacd_aac_imdct.h"
#include "ixheaacd_pulsedata.h"
#include "ixheaacd_drc_data_struct.h"
#include "ixheaacd_lt_predict.h"
#include "ixheaacd_cnst.h"
#include "ixheaacd_ec_defines.h"
#include "ixheaacd_ec_struct_def.h"
#include "ixheaacd_channelinfo.h"
#include "ixheaacd_drc_dec.h"
#include "ixheaacd_sbrdecoder.h"
#include "ixheaacd_tns.h"
#include "ixheaacd_sbr_scale.h"
#include "ixheaacd_lpp_tran.h"
#include "ixheaacd_env_extr_part.h"
#include "ixheaacd_sbr_rom.h"
#include "ixheaacd_block.h"
#include "ixheaacd_hybrid.h"
#include "ixheaacd_ps_dec.h"
#include "ixheaacd_env_extr.h"
#include "ixheaacd_basic_funcs.h"
#include "ixheaacd_env_calc.h"
#include "ixheaacd_audioobjtypes.h"

/* One channel, one frame through the reference's ixheaacd_imdct_process
 * (decoder/ixheaacd_lpfuncs.c:347).
 *   spec[1024]      in, clobbered (the reference works in place)
 *   overlap[512]    in/out  (ia_aac_dec_overlap_info.ptr_overlap_buf)
 *   prev_seq/shape  in/out  (overlap_info.window_sequence / window_shape)
 *   out[1024*ch_fac] written at stride ch_fac
 * returns ics.qshift_adj */
int ref_imdct_process(WORD32 *spec, WORD32 *overlap, WORD16 *prev_seq, WORD16 *prev_shape,
                      int seq, int shape, WORD32 *out, int ch_fac) {
  static __thread WORD32 scratch[2048];
  ia_aac_dec_overlap_info oi;
  ia_ics_info_struct ics;
  ia_aac_dec_tables_struct tabs;
  ia_aac_dec_imdct_tables_struct *rom = (ia_aac_dec_imdct_tables_struct *)&ixheaacd_imdct_tables;
  memset(&oi, 0, sizeof(oi));
  memset(&ics, 0, sizeof(ics));
  memset(&tabs, 0, sizeof(tabs));
  tabs.pstr_imdct_tables = rom;
  oi.ptr_long_window[0] = rom->only_long_window_sine;
  oi.ptr_short_window[0] = rom->only_short_window_sine;
  oi.ptr_long_window[1] = rom->only_long_window_kbd;
  oi.ptr_short_window[1] = rom->only_short_window_kbd;
  oi.window_shape = *prev_shape;
  oi.window_sequence = *prev_seq;
  oi.ptr_overlap_buf = overlap;
  ics.window_sequence = (WORD16)seq;
  ics.window_shape = (WORD16)shape;
  ics.frame_length = 1024;
  ixheaacd_imdct_process(&oi, spec, &ics, out, (WORD16)ch_fac, scratch, &tabs, AOT_AAC_LC, 0, 0);
  *prev_seq = oi.window_sequence;
  *prev_shape = oi.window_shape;
  return ics.qshift_adj;
}

/* The same for frame_length 960 (lpfuncs.c's 960 branches: ixheaacd_mdct_960 / ixheaacd_inverse_transform_960, the 960- and
 * 120-sample windows as aacdecoder.c:210-227 selects them): spec[1024] (960 used, clobbered), overlap[512] (480 used),
 * out[960 * ch_fac]. */
int ref_imdct960_process(WORD32 *spec, WORD32 *overlap, WORD16 *prev_seq, WORD16 *prev_shape, int seq, int shape,
                         WORD32 *out, int ch_fac) {
  static __thread WORD32 scratch[2048];
  ia_aac_dec_overlap_info oi;
  ia_ics_info_struct ics;
  ia_aac_dec_tables_struct tabs;
  ia_aac_dec_imdct_tables_struct *rom = (ia_aac_dec_imdct_tables_struct *)&ixheaacd_imdct_tables;
  memset(&oi, 0, sizeof(oi));
  memset(&ics, 0, sizeof(ics));
  memset(&tabs, 0, sizeof(tabs));
  tabs.pstr_imdct_tables = rom;
  oi.ptr_long_window[0] = rom->only_long_window_sine_960;
  oi.ptr_short_window[0] = rom->only_short_window_sine_120;
  oi.ptr_long_window[1] = rom->only_long_window_kbd_960;
  oi.ptr_short_window[1] = rom->only_short_window_kbd_120;
  oi.window_shape = *prev_shape;
  oi.window_sequence = *prev_seq;
  oi.ptr_overlap_buf = overlap;
  ics.window_sequence = (WORD16)seq;
  ics.window_shape = (WORD16)shape;
  ics.frame_length = 960;
  ixheaacd_imdct_process(&oi, spec, &ics, out, (WORD16)ch_fac, scratch, &tabs, AOT_AAC_LC, 0, 0);
  *prev_seq = oi.window_sequence;
  *prev_shape = oi.window_shape;
  return ics.qshift_adj;
}

/* AAC-LD / ELD: frame_length 512 or 480, object_type AOT_ER_AAC_LD (23) or AOT_ER_AAC_ELD (39), ONLY_LONG frames; windows as
 * aacdecoder.c:236-277 selects them.  spec[2048] (frame_length lines in, clobbered: ELD works in 4 x frame_length words),
 * overlap[2048] in/out (LD: frame_length / 2 words, ELD: 3 x frame_length), out16[frame_length * ch_fac] PCM16. */
int ref_imdct_ld_process(WORD32 *spec, WORD32 *overlap, WORD16 *prev_shape, int shape, int frame_length, int aot,
                         WORD16 *out16, int ch_fac) {
  static __thread WORD32 scratch[4096];
  ia_aac_dec_overlap_info oi;
  ia_ics_info_struct ics;
  ia_aac_dec_tables_struct tabs;
  ia_aac_dec_imdct_tables_struct *rom = (ia_aac_dec_imdct_tables_struct *)&ixheaacd_imdct_tables;
  memset(&oi, 0, sizeof(oi));
  memset(&ics, 0, sizeof(ics));
  memset(&tabs, 0, sizeof(tabs));
  tabs.pstr_imdct_tables = rom;
  if (aot == AOT_ER_AAC_ELD) {
    oi.ptr_long_window[0] = oi.ptr_long_window[1] =
        frame_length == 512 ? (WORD16 *)rom->window_sine_512_eld : (WORD16 *)rom->window_sine_480_eld;
  } else {
    oi.ptr_long_window[0] = frame_length == 512 ? (WORD16 *)rom->window_sine_512 : (WORD16 *)rom->window_sine_480;
    oi.ptr_long_window[1] = frame_length == 512 ? (WORD16 *)rom->low_overlap_win : (WORD16 *)rom->low_overlap_win_480;
  }
  oi.ptr_short_window[0] = rom->only_short_window_sine;
  oi.ptr_short_window[1] = rom->only_short_window_kbd;
  oi.window_shape = *prev_shape;
  oi.window_sequence = 0;
  oi.ptr_overlap_buf = overlap;
  ics.window_sequence = 0;
  ics.window_shape = (WORD16)shape;
  ics.frame_length = (WORD16)frame_length;
  ixheaacd_imdct_process(&oi, spec, &ics, out16, (WORD16)ch_fac, scratch, &tabs, aot, 0, 0);
  *prev_shape = oi.window_shape;
  return ics.qshift_adj;
}

/* C loops over n channel-frames for timing the reference as the CPU baseline of the 960-line and LD / ELD rows
 * (ONLY_LONG frames; buffers at the reference's own strides: spec 2048 words, overlap 2048 words per channel) */
void ref_imdct960_loop(int n, WORD32 *spec, WORD32 *overlap, WORD32 *out) {
  int c;
  for (c = 0; c < n; c++) {
    WORD16 ps = 0, pw = 0;
    ref_imdct960_process(spec + 2048 * (size_t)c, overlap + 2048 * (size_t)c, &ps, &pw, 0, 0, out, 1);
  }
}
void ref_imdct_ld_loop(int n, WORD32 *spec, WORD32 *overlap, int frame_length, int aot, WORD16 *out16) {
  int c;
  for (c = 0; c < n; c++) {
    WORD16 pw = 0;
    ref_imdct_ld_process(spec + 2048 * (size_t)c, overlap + 2048 * (size_t)c, &pw, 0, frame_length, aot, out16, 1);
  }
}

/* n channel-frames in a C loop (for timing the reference as the CPU baseline):
 * per channel c: spec[c][1024] (clobbered), overlap[c][512], prev_seq/prev_shape[c],
 * seq/shape[c]; writes PCM16 at stride 1 per channel like ixheaacd_scale_adjust +
 * round16 would (AAC-LC, limiter off).  Thread-safe across disjoint channel ranges. */
void ref_imdct_batch(int n, WORD32 *spec, WORD32 *overlap, WORD16 *prev_seq, WORD16 *prev_shape,
                     const UWORD8 *seq, const UWORD8 *shape, WORD16 *pcm) {
  WORD32 out[1024];
  int c, i;
  for (c = 0; c < n; c++) {
    int q = ref_imdct_process(spec + 1024 * (size_t)c, overlap + 512 * (size_t)c, prev_seq + c, prev_shape + c,
                              seq[c], shape[c], out, 1);
    if (pcm)
      for (i = 0; i < 1024; i++) pcm[1024 * (size_t)c + i] = ixheaac_round16((WORD32)((UWORD32)out[i] << q));
  }
}

/* ---- the WORD32 -> WORD16 hand-off behind the IMDCT, by the reference's own code ------------
 * mode 1 (core -> SBR): ixheaacd_allocate_sbr_scr (api.c:337-370) converts the interleaved WORD32 block in place:
 *        round16(shl32_sat(x, qshift_adj[ch])), the WORD16 samples packed at the front of the same buffer;
 * mode 0 (AAC-LC, limiter off): ixheaacd_scale_adjust (peak_limiter.c:324), then the round16 loop of api.c:3676-3681
 *        (that loop is inline in ixheaacd_dec_execute: it is restated here with the reference's ixheaac_round16).
 * x: [1024][nch] interleaved WORD32 (clobbered), q: [nch], out: [1024][nch] WORD16. */
/* (ia_sbr_scr_struct: ixheaacd_sbrdecoder.h, included above) */
VOID ixheaacd_allocate_sbr_scr(ia_sbr_scr_struct *sbr_scratch_struct, VOID *base_scratch_ptr, VOID *output_ptr,
                               WORD32 total_channels, WORD8 *p_qshift_arr, UWORD8 slot_pos, UWORD8 num_ch);
VOID ixheaacd_scale_adjust(WORD32 *samples, UWORD32 frame_len, WORD8 *qshift_adj, WORD num_channels);

void ref_pcm_handoff(WORD32 *x, WORD8 *q, int nch, int mode, WORD16 *out) {
  int i;
  if (mode == 1) {
    ia_sbr_scr_struct scr;
    static __thread WORD32 scratch[16];
    ixheaacd_allocate_sbr_scr(&scr, scratch, x, nch, q, 0, (UWORD8)nch);
    memcpy(out, x, sizeof(WORD16) * 1024 * (size_t)nch);
  } else {
    ixheaacd_scale_adjust(x, 1024, q, nch);
    for (i = 0; i < 1024 * nch; i++) out[i] = ixheaac_round16(x[i]);
  }
}

/* ======================================================================================
 * SBR QMF banks (fixed-point Path B).  Wrappers fill the reference's own structs and call
 * its external symbols; tables are the reference's ROM (ixheaacd_aac_qmf_dec_tables).
 * ====================================================================================== */
#include "ixheaacd_qmf_dec.h"

static ia_qmf_dec_tables_struct *ref_qmf_tabs(void) {
  return (ia_qmf_dec_tables_struct *)&ixheaacd_aac_qmf_dec_tables;
}

void ref_radix4bfly(int which_w, int w_off, WORD32 *x, int index1, int index) {
  const WORD16 *w = which_w == 32 ? ref_qmf_tabs()->w_32 : ref_qmf_tabs()->w_16;
  ixheaacd_radix4bfly(w + w_off, x, index1, index);
}
void ref_postradix4(WORD32 *y, WORD32 *x) { ixheaacd_postradixcompute4(y, x, ref_qmf_tabs()->dig_rev_table4_16, 16); }
void ref_postradix2(WORD32 *y, WORD32 *x) { ixheaacd_postradixcompute2(y, x, ref_qmf_tabs()->dig_rev_table2_32, 32); }
void ref_dct3_32(WORD32 *in, WORD32 *out) {
  ia_qmf_dec_tables_struct *t = ref_qmf_tabs();
  ixheaacd_dct3_32(in, out, t->dct23_tw, t->post_fft_tbl, t->w_16, t->dig_rev_table4_16);
}
/* cos_sin_mod with the twiddle set the analysis (m=16) / synthesis (m=32) bank selects */
void ref_cos_sin_mod(WORD32 *s, int m) {
  ia_qmf_dec_tables_struct *t = ref_qmf_tabs();
  ia_sbr_qmf_filter_bank_struct bank;
  memset(&bank, 0, sizeof(bank));
  bank.no_channels = 2 * m;
  if (m == 32) {
    bank.cos_twiddle = t->sbr_sin_cos_twiddle_l64;
    bank.alt_sin_twiddle = t->sbr_alt_sin_twiddle_l64;
    ixheaacd_cos_sin_mod(s, &bank, t->w_32, t->dig_rev_table2_32);
  } else {
    bank.cos_twiddle = t->sbr_sin_cos_twiddle_l32;
    bank.alt_sin_twiddle = t->sbr_alt_sin_twiddle_l32;
    ixheaacd_cos_sin_mod(s, &bank, t->w_16, t->dig_rev_table4_16);
  }
}
/* LP synthesis slot: ixheaacd_inv_modulation_lp writes 128 samples at filter_states */
void ref_inv_modulation_lp(WORD32 *x, WORD16 *b) {
  ia_sbr_qmf_filter_bank_struct bank;
  memset(&bank, 0, sizeof(bank));
  bank.no_channels = 64;
  ixheaacd_inv_modulation_lp(x, b, &bank, ref_qmf_tabs());
}
/* HQ synthesis slot: inv_emodulation + shiftrountine_with_rnd */
void ref_synth_hq_slot(WORD32 *s, WORD16 *b, int shift) {
  ia_qmf_dec_tables_struct *t = ref_qmf_tabs();
  ia_sbr_qmf_filter_bank_struct bank;
  memset(&bank, 0, sizeof(bank));
  bank.no_channels = 64;
  bank.cos_twiddle = t->sbr_sin_cos_twiddle_l64;
  bank.alt_sin_twiddle = t->sbr_alt_sin_twiddle_l64;
  ixheaacd_inv_emodulation(s, &bank, t);
  ixheaacd_shiftrountine_with_rnd(s, s + 64, b, 64, shift);
}

/* One frame through ixheaacd_cplx_anal_qmffilt.  ring/wr/phase = anal_filter_states,
 * core_samples_buffer - anal_filter_states, filter_pos - qmf_c.  qmf: slot s at qmf + s*slot_stride
 * (real at +0, imaginary at +64 in HQ mode). */
void ref_qmf_analysis(const WORD16 *pcm, int stride, WORD16 *ring, WORD16 *wr, WORD16 *phase, int low_pow,
                      int usb, WORD32 *qmf, int slot_stride) {
  ia_qmf_dec_tables_struct *t = ref_qmf_tabs();
  ia_sbr_qmf_filter_bank_struct bank;
  ia_sbr_scale_fact_struct sf;
  WORD32 *re[32], *im[32];
  int s;
  memset(&bank, 0, sizeof(bank));
  memset(&sf, 0, sizeof(sf));
  bank.no_channels = 32;
  bank.num_time_slots = 32;
  bank.lsb = 0;
  bank.usb = (WORD16)usb;
  bank.anal_filter_states = ring;
  bank.core_samples_buffer = ring + *wr;
  bank.analy_win_coeff = t->qmf_c;
  bank.filter_pos = t->qmf_c + *phase;
  for (s = 0; s < 32; s++) {
    re[s] = qmf + (size_t)s * slot_stride;
    im[s] = re[s] + 64;
  }
  ixheaacd_cplx_anal_qmffilt(pcm, &sf, re, im, &bank, t, stride, low_pow, AOT_AAC_LC);
  *wr = (WORD16)(bank.core_samples_buffer - ring);
  *phase = (WORD16)(bank.filter_pos - t->qmf_c);
}

/* the LD / ELD flavour: state4 = {core_samples_buffer, filter_pos, filter_2, fp1_anal} as offsets; a new stream starts
   with {0, 0, 32, 0} (sbrdec_initfuncs.c:1122-1148) */
void ref_qmf_analysis_eld(const WORD16 *pcm, int stride, WORD16 *ring, WORD16 *state4, int n_slots, int usb, WORD32 *qmf,
                          int slot_stride) {
  ia_qmf_dec_tables_struct *t = ref_qmf_tabs();
  ia_sbr_qmf_filter_bank_struct bank;
  ia_sbr_scale_fact_struct sf;
  WORD32 *re[32], *im[32];
  int s;
  memset(&bank, 0, sizeof(bank));
  memset(&sf, 0, sizeof(sf));
  bank.no_channels = 32;
  bank.num_time_slots = (WORD16)n_slots;
  bank.lsb = 0;
  bank.usb = (WORD16)usb;
  bank.anal_filter_states = ring;
  bank.core_samples_buffer = ring + state4[0];
  bank.analy_win_coeff = t->qmf_c_eld3;
  bank.filter_pos = t->qmf_c_eld3 + state4[1];
  bank.filter_2 = t->qmf_c_eld3 + state4[2];
  bank.fp1_anal = ring + state4[3];
  bank.fp2_anal = ring + (32 - state4[3]);
  for (s = 0; s < 32; s++) {
    re[s] = qmf + (size_t)s * slot_stride;
    im[s] = re[s] + 64;
  }
  ixheaacd_cplx_anal_qmffilt(pcm, &sf, re, im, &bank, t, stride, 0, AOT_ER_AAC_ELD);
  state4[0] = (WORD16)(bank.core_samples_buffer - ring);
  state4[1] = (WORD16)(bank.filter_pos - t->qmf_c_eld3);
  state4[2] = (WORD16)(bank.filter_2 - t->qmf_c_eld3);
  state4[3] = (WORD16)(bank.fp1_anal - ring);
}

/* One frame through ixheaacd_cplx_synt_qmffilt (no PS: active = 0).  qmf is scaled and transformed
 * in place by the reference.  sf = {lb_scale, ov_lb_scale, hb_scale, st_syn_scale}. */
void ref_qmf_synthesis(WORD32 *qmf, int slot_stride, const WORD16 *sfv, int lsb, int usb, int split, WORD16 *ring,
                       WORD16 *drc_offset, WORD16 *phase, int low_pow, WORD16 *pcm, int stride) {
  ia_qmf_dec_tables_struct *t = ref_qmf_tabs();
  ia_sbr_tables_struct tabs;
  ia_sbr_qmf_filter_bank_struct bank;
  ia_sbr_scale_fact_struct sf;
  static __thread WORD32 copy_re[32][64], copy_im[32][64];
  WORD32 *re[MAX_ENV_COLS], *im[MAX_ENV_COLS], *ore[MAX_ENV_COLS], *oim[MAX_ENV_COLS];
  int s;
  memset(&bank, 0, sizeof(bank));
  memset(&sf, 0, sizeof(sf));
  memset(&tabs, 0, sizeof(tabs));
  tabs.qmf_dec_tables_ptr = t;
  sf.lb_scale = sfv[0];
  sf.ov_lb_scale = sfv[1];
  sf.hb_scale = sfv[2];
  sf.st_syn_scale = sfv[3];
  bank.no_channels = 64;
  bank.num_time_slots = 32;
  bank.lsb = (WORD16)lsb;
  bank.usb = (WORD16)usb;
  bank.filter_states = ring;
  bank.p_filter = t->qmf_c;
  bank.filter_pos_syn = t->qmf_c + *phase;
  bank.ixheaacd_drc_offset = *drc_offset;
  for (s = 0; s < 32; s++) {
    re[s] = qmf + (size_t)s * slot_stride;
    im[s] = re[s] + 64;
    ore[s] = copy_re[s];
    oim[s] = copy_im[s];
  }
  ixheaacd_cplx_synt_qmffilt(re, im, split, ore, oim, &sf, pcm, &bank, NULL, 0, low_pow, &tabs, NULL, stride, 0, NULL,
                             AOT_AAC_LC);
  *drc_offset = bank.ixheaacd_drc_offset;
  *phase = (WORD16)(bank.filter_pos_syn - t->qmf_c);
}

/* the LD / ELD flavour (AOT_ER_AAC_ELD): state4 = {ixheaacd_drc_offset, filter_pos_syn - qmf_c_eld, fp1_syn - ring, sixty4};
   a new stream: {0, 0, 0, 64} (sbrdec_initfuncs.c:1181-1209).  qmf is scaled and transformed in place by the reference. */
/* the rows the LD / ELD synthesis bank hands on through qmf_real_out / qmf_imag_out (qmf_dec.c:966-976): the region-rescaled
   matrix (its in-place input rows are work space after the call) */
static __thread WORD32 eld_out_re[32][64], eld_out_im[32][64];
void ref_qmf_synthesis_eld_handed_on(WORD32 *dst, int n_slots, int slot_stride) {
  int s, k;
  for (s = 0; s < n_slots; s++)
    for (k = 0; k < 64; k++) {
      dst[(size_t)s * slot_stride + k] = eld_out_re[s][k];
      dst[(size_t)s * slot_stride + 64 + k] = eld_out_im[s][k];
    }
}

void ref_qmf_synthesis_eld(WORD32 *qmf, int slot_stride, const WORD16 *sfv, int lsb, int usb, int split, WORD16 *ring,
                           WORD16 *state4, int n_slots, WORD16 *pcm, int stride) {
  ia_qmf_dec_tables_struct *t = ref_qmf_tabs();
  ia_sbr_tables_struct tabs;
  ia_sbr_qmf_filter_bank_struct bank;
  ia_sbr_scale_fact_struct sf;
  WORD32 *re[MAX_ENV_COLS], *im[MAX_ENV_COLS], *ore[MAX_ENV_COLS], *oim[MAX_ENV_COLS];
  int s;
  memset(&bank, 0, sizeof(bank));
  memset(&sf, 0, sizeof(sf));
  memset(&tabs, 0, sizeof(tabs));
  tabs.qmf_dec_tables_ptr = t;
  sf.lb_scale = sfv[0];
  sf.ov_lb_scale = sfv[1];
  sf.hb_scale = sfv[2];
  sf.st_syn_scale = sfv[3];
  bank.no_channels = 64;
  bank.num_time_slots = (WORD16)n_slots;
  bank.lsb = (WORD16)lsb;
  bank.usb = (WORD16)usb;
  bank.filter_states = ring;
  bank.p_filter = t->qmf_c_eld;
  bank.filter_pos_syn = t->qmf_c_eld + state4[1];
  bank.ixheaacd_drc_offset = state4[0];
  bank.fp1_syn = ring + state4[2];
  bank.sixty4 = state4[3];
  bank.fp2_syn = bank.fp1_syn + bank.sixty4;
  for (s = 0; s < 32; s++) {
    re[s] = qmf + (size_t)s * slot_stride;
    im[s] = re[s] + 64;
    ore[s] = eld_out_re[s];
    oim[s] = eld_out_im[s];
  }
  ixheaacd_cplx_synt_qmffilt(re, im, split, ore, oim, &sf, pcm, &bank, NULL, 0, 0, &tabs, NULL, stride, 0, NULL,
                             AOT_ER_AAC_ELD);
  state4[0] = bank.ixheaacd_drc_offset;
  state4[1] = (WORD16)(bank.filter_pos_syn - t->qmf_c_eld);
  state4[2] = (WORD16)(bank.fp1_syn - ring);
  state4[3] = (WORD16)bank.sixty4;
}
