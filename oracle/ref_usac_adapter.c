/*
 * oracle/ref_usac_adapter.c -- TEST INFRASTRUCTURE ONLY.
 * Drives the real ixheaacd_fd_frm_dec (decoder/ixheaacd_imdct.c:596) for one channel-frame: fills the reference's own
 * ia_usac_data_struct (the members the function reads: ixheaacd_main.h:69-191) and calls the reference's symbol.
 * Contains no reference code.  The header list is what the struct's definition needs.
 */
#include <stdlib.h>
#include <string.h>

#include "ixheaac_type_def.h"
#include "ixheaacd_interface.h"
#include "ixheaacd_defines.h"
#include "ixheaacd_aac_rom.h"
#include "ixheaacd_bitbuffer.h"
#include "ixheaacd_tns_usac.h"
#include "ixheaacd_cnst.h"
#include "ixheaacd_acelp_info.h"
#include "ixheaacd_td_mdct.h"
#include "ixheaacd_sbrdecsettings.h"
#include "ixheaacd_info.h"
#include "ixheaacd_sbr_common.h"
#include "ixheaacd_drc_data_struct.h"
#include "ixheaacd_drc_dec.h"
#include "ixheaacd_sbrdecoder.h"
#include "ixheaacd_mps_polyphase.h"
#include "ixheaac_sbr_const.h"
#include "ixheaacd_pulsedata.h"
#include "ixheaacd_pns.h"
#include "ixheaacd_lt_predict.h"
#include "ixheaacd_ec_defines.h"
#include "ixheaacd_ec_struct_def.h"
#include "ixheaacd_main.h"

WORD32 ixheaacd_fd_frm_dec(ia_usac_data_struct *usac_data, WORD32 i_ch);

/* coef: ccfl lines in / out (the reference transforms coef_fix in place); overlap: ccfl words in / out; out: ccfl Q15
   words; time: ccfl floats as the caller makes them (ixheaacd_ext_ch_ele.c:1008-1012).  ccfl 1024 or 768. */
int ref_usac_fd_imdct_ccfl(WORD32 *coef, WORD32 *overlap, int ccfl, int seq, int shape, int shape_prev, WORD32 *out, FLOAT32 *time) {
  static __thread ia_usac_data_struct *u;
  int k, err;
  if (!u) u = (ia_usac_data_struct *)calloc(1, sizeof(*u));
  u->ccfl = ccfl;
  u->ec_flag = 0;
  u->frame_ok = 1;
  u->td_frame_prev[0] = 0;
  u->fac_data_present[0] = 0;
  u->window_sequence[0] = seq;
  u->window_shape[0] = shape;
  u->window_shape_prev[0] = shape_prev;
  u->coef_fix[0] = u->arr_coef_fix[0];
  u->str_tddec[0] = &u->arr_str_tddec[0];
  memcpy(u->coef_fix[0], coef, sizeof(WORD32) * ccfl);
  memcpy(u->overlap_data_ptr[0], overlap, sizeof(WORD32) * ccfl);
  memset(u->output_data_ptr[0], 0, sizeof(WORD32) * ccfl);
  err = ixheaacd_fd_frm_dec(u, 0);
  memcpy(coef, u->coef_fix[0], sizeof(WORD32) * ccfl);
  memcpy(overlap, u->overlap_data_ptr[0], sizeof(WORD32) * ccfl);
  memcpy(out, u->output_data_ptr[0], sizeof(WORD32) * ccfl);
  for (k = 0; k < ccfl; k++) time[k] = (FLOAT32)((FLOAT32)out[k] * (FLOAT32)0.000030517578125) /* ONE_BY_TWO_POW_15, ext_ch_ele.c:146 */;
  return err;
}
int ref_usac_fd_imdct(WORD32 *coef, WORD32 *overlap, int seq, int shape, int shape_prev, WORD32 *out, FLOAT32 *time) {
  return ref_usac_fd_imdct_ccfl(coef, overlap, 1024, seq, shape, shape_prev, out, time);
}

/* The frame behind an LPD frame and / or with FAC data.  The FAC signal is the reference's own: ixheaacd_cal_fac_data
   (imdct.c:210) runs inside ixheaacd_fd_frm_dec on the LPD-side inputs given here (fac_data: gain index + lfac quantised
   lines; lpc_prev: ORDER + 1 coefficients; acelp_in: the ACELP zero-input response) and, before that, once more on copies
   so that fac_out (2 lfac words as the windowing reads them) and fac_q can be handed back -- they are what the boundary
   takes from the host.  The LPD decoder's bass post filter (ixheaacd_lpd_bpf_fix, LPD state) is linked out as the identity
   (Makefile.ref: --wrap): the boundary leaves it to the LPD decoder's side. */
IA_ERRORCODE ixheaacd_cal_fac_data(ia_usac_data_struct *usac_data, WORD32 i_ch, WORD32 n_long, WORD32 lfac, WORD32 *fac_idata,
                                   WORD8 *q_fac);
WORD32 __wrap_ixheaacd_lpd_bpf_fix(ia_usac_data_struct *usac_data, WORD32 is_short_flag, FLOAT32 out_buffer[], VOID *st) {
  (void)usac_data; (void)is_short_flag; (void)out_buffer; (void)st;
  return 0;
}
int ref_usac_fd_imdct_lpd(WORD32 *coef, WORD32 *overlap, int ccfl, int seq, int shape, int shape_prev, int td_prev, int fac_present,
                          const WORD32 *fac_data, const FLOAT32 *lpc_prev, const FLOAT32 *acelp_in, WORD32 *out, WORD32 *fac_out,
                          WORD32 *fac_q_out) {
  static __thread ia_usac_data_struct *u;
  int err, lfac;
  if (!u) u = (ia_usac_data_struct *)calloc(1, sizeof(*u));
  u->ccfl = ccfl;
  u->ec_flag = 0;
  u->frame_ok = 1;
  u->num_subfrm = ccfl == 768 ? 3 : 4;
  u->td_frame_prev[0] = td_prev;
  u->fac_data_present[0] = fac_present;
  u->window_sequence[0] = seq;
  u->window_shape[0] = shape;
  u->window_shape_prev[0] = shape_prev;
  u->coef_fix[0] = u->arr_coef_fix[0];
  u->str_tddec[0] = &u->arr_str_tddec[0];
  lfac = td_prev ? (seq == 2 ? ccfl >> 4 : ccfl >> 3) : FAC_LENGTH;
  if (fac_present) {
    WORD32 tmp[2 * FAC_LENGTH + 16];
    WORD8 q = 0;
    memcpy(u->lpc_prev[0], lpc_prev, sizeof(u->lpc_prev[0]));
    memcpy(u->acelp_in[0], acelp_in, sizeof(u->acelp_in[0]));
    memcpy(u->fac_data[0], fac_data, sizeof(WORD32) * (FAC_LENGTH + 1));
    memset(tmp, 0, sizeof(tmp));
    err = ixheaacd_cal_fac_data(u, 0, ccfl, lfac, tmp, &q);
    if (err) return err;
    memcpy(fac_out, tmp, sizeof(WORD32) * 2 * lfac);
    *fac_q_out = q;
    memcpy(u->fac_data[0], fac_data, sizeof(WORD32) * (FAC_LENGTH + 1)); /* cal_fac_data scales it in place */
  }
  memcpy(u->coef_fix[0], coef, sizeof(WORD32) * ccfl);
  memcpy(u->overlap_data_ptr[0], overlap, sizeof(WORD32) * ccfl);
  memset(u->output_data_ptr[0], 0, sizeof(WORD32) * ccfl);
  err = ixheaacd_fd_frm_dec(u, 0);
  memcpy(coef, u->coef_fix[0], sizeof(WORD32) * ccfl);
  memcpy(overlap, u->overlap_data_ptr[0], sizeof(WORD32) * ccfl);
  memcpy(out, u->output_data_ptr[0], sizeof(WORD32) * ccfl);
  return err;
}

/* the general FFT alone: ixheaacd_complex_fft (fft.c:2664) with fft_mode = -1; returns what it leaves in *preshift (0 on entry) */
VOID ixheaacd_complex_fft(WORD32 *data_r, WORD32 *data_i, WORD32 nlength, WORD32 fft_mode, WORD32 *preshift);
int ref_fft_fwd(WORD32 *xr, WORD32 *xi, int n) {
  WORD32 pre = 0;
  ixheaacd_complex_fft(xr, xi, n, -1, &pre);
  return pre;
}
