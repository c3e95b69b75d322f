/*
 * oracle/ref_dropin_usac.c -- TEST INFRASTRUCTURE ONLY, part of oracle/_ref/xaacdec_dropin (Makefile.ref: --wrap=ixheaacd_fd_frm_dec).
 *
 * The USAC frequency-domain seam of the drop-in: ixheaacd_fd_frm_dec (decoder/ixheaacd_imdct.c:596, call sites
 * ixheaacd_ext_ch_ele.c:799 / :991) served by xaac_usac_imdct_process_batch -- the stub INTEGRATION.md section 7 describes, as
 * code.  The forward-aliasing-cancellation signal of a frame behind an LPD frame (ixheaacd_cal_fac_data, imdct.c:210) is made on
 * the device from the LPD decoder's three inputs -- FAC lines, previous LPC filter, ACELP zero-input response: xaac_usac_fac_in --
 * ($XAAC_DROPIN_HOST_FAC: by the reference's own function on the host, the signal handed over instead); the LPD decoder's bass
 * post filter (ixheaacd_lpd_bpf_fix, lpc.c:749) stays the reference's and runs behind the call where it ran.  Contains no reference code; the
 * header list is what ia_usac_data_struct's definition needs (as in ref_usac_adapter.c).
 */
#include <stdio.h>
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

#include <hip/hip_runtime_api.h>
#include "xaac_amd.h"

xaac_ctx *dropin_ctx(void);        /* ref_dropin.c: the drop-in's context (created, and the summary registered, on first use) */
void dropin_count_usac_imdct(int with_fac, int behind_lpd);
void dropin_count_usac_fac_on_device(void);

WORD32 __real_ixheaacd_fd_frm_dec(ia_usac_data_struct *usac_data, WORD32 i_ch);
IA_ERRORCODE ixheaacd_cal_fac_data(ia_usac_data_struct *usac_data, WORD32 i_ch, WORD32 n_long, WORD32 lfac, WORD32 *fac_idata,
                                   WORD8 *q_fac);

#define HIPU(x) do { if ((x) != hipSuccess) { fprintf(stderr, "xaacdec_dropin: %s failed\n", #x); exit(3); } } while (0)

WORD32 __wrap_ixheaacd_fd_frm_dec(ia_usac_data_struct *u, WORD32 ch) {
  static struct { int32_t *coef, *overlap, *out32, *status; xaac_usac_ics *ics; uint8_t *shape_prev, *flags; xaac_usac_fac *fac; xaac_usac_fac_in *fac_in; } d;
  static xaac_usac_fac_in fin;
  const int host_fac = getenv("XAAC_DROPIN_HOST_FAC") != NULL;
  const int ccfl = u->ccfl, seq = u->window_sequence[ch];
  const int td_prev = u->td_frame_prev[ch] != 0, fac_apply = u->fac_data_present[ch] && u->frame_ok == 1;
  xaac_usac_imdct_batch b;
  xaac_usac_ics ics;
  static xaac_usac_fac fac;
  uint8_t flags, shape_prev;
  int32_t status = 0;
  xaac_ctx *ctx;
  int k;
  if (getenv("XAAC_DROPIN_PASS_USAC_IMDCT") || u->ec_flag || (ccfl != 1024 && ccfl != 768)) return __real_ixheaacd_fd_frm_dec(u, ch);
  ctx = dropin_ctx();
  if (!d.coef) {
    HIPU(hipMalloc((void **)&d.coef, 1024 * 4));
    HIPU(hipMalloc((void **)&d.overlap, 1024 * 4));
    HIPU(hipMalloc((void **)&d.out32, 1024 * 4));
    HIPU(hipMalloc((void **)&d.status, 4));
    HIPU(hipMalloc((void **)&d.ics, sizeof(xaac_usac_ics)));
    HIPU(hipMalloc((void **)&d.shape_prev, 1));
    HIPU(hipMalloc((void **)&d.flags, 1));
    HIPU(hipMalloc((void **)&d.fac, sizeof(xaac_usac_fac)));
    HIPU(hipMalloc((void **)&d.fac_in, sizeof(xaac_usac_fac_in)));
  }
  memset(&fac, 0, sizeof(fac));
  if (fac_apply && td_prev && !host_fac) { /* the function's inputs as the LPD decoder left them (imdct.c:226-229); it runs on the device */
    memset(&fin, 0, sizeof(fin));
    memcpy(fin.fac_data, u->fac_data[ch], sizeof(fin.fac_data));
    memcpy(fin.lpc_prev, u->lpc_prev[ch], sizeof(fin.lpc_prev));
    memcpy(fin.acelp_in, u->acelp_in[ch], sizeof(float) * (size_t)(ccfl / 4));
    HIPU(hipMemcpy(d.fac_in, &fin, sizeof(fin), hipMemcpyHostToDevice));
  } else if (fac_apply) { /* the LPD side's own code, unchanged (imdct.c:618-640) */
    WORD32 fac_idata[2 * FAC_LENGTH + 16];
    WORD8 q = 0;
    const int lfac = td_prev ? (seq == EIGHT_SHORT_SEQUENCE ? ccfl >> 4 : ccfl >> 3) : FAC_LENGTH;
    IA_ERRORCODE err;
    memset(fac_idata, 0, sizeof(fac_idata));
    err = ixheaacd_cal_fac_data(u, ch, ccfl, lfac, fac_idata, &q);
    if (err) return err;
    fac.q = q;
    memcpy(fac.data, fac_idata, sizeof(WORD32) * 2 * (lfac < FAC_LENGTH ? lfac : FAC_LENGTH));
  }
  ics.window_sequence = (uint8_t)seq;
  ics.window_shape = (uint8_t)u->window_shape[ch];
  shape_prev = (uint8_t)u->window_shape_prev[ch];
  flags = (uint8_t)((td_prev ? 1 : 0) | (fac_apply ? 2 : 0));
  HIPU(hipMemcpy(d.coef, u->coef_fix[ch], (size_t)ccfl * 4, hipMemcpyHostToDevice));
  HIPU(hipMemcpy(d.overlap, u->overlap_data_ptr[ch], (size_t)ccfl * 4, hipMemcpyHostToDevice));
  HIPU(hipMemcpy(d.ics, &ics, sizeof(ics), hipMemcpyHostToDevice));
  HIPU(hipMemcpy(d.shape_prev, &shape_prev, 1, hipMemcpyHostToDevice));
  HIPU(hipMemcpy(d.flags, &flags, 1, hipMemcpyHostToDevice));
  HIPU(hipMemcpy(d.fac, &fac, sizeof(fac), hipMemcpyHostToDevice));
  memset(&b, 0, sizeof(b));
  b.n_ch = 1;
  b.ccfl = ccfl;
  b.coef = d.coef;
  b.ics = d.ics;
  b.overlap = d.overlap;
  b.shape_prev = d.shape_prev;
  b.out32 = d.out32;
  b.status = d.status;
  b.lpd_flags = d.flags;
  b.fac = d.fac;
  if (fac_apply && td_prev && !host_fac) {
    b.fac_in = d.fac_in;
    b.fac_work = d.fac;
  }
  if (xaac_usac_imdct_process_batch(ctx, &b) != XAAC_OK || xaac_sync(ctx) != XAAC_OK) {
    fprintf(stderr, "xaacdec_dropin: xaac_usac_imdct_process_batch failed\n");
    exit(3);
  }
  HIPU(hipMemcpy(&status, d.status, 4, hipMemcpyDeviceToHost));
  if (status) { /* a frame the boundary refuses (fac_q outside the windowing's shift range): the reference's own code */
    if (getenv("XAAC_DROPIN_DEBUG")) fprintf(stderr, "xaacdec_dropin: USAC FD frame refused (%d): the reference's own ixheaacd_fd_imdct\n", (int)status);
    /* (ixheaacd_cal_fac_data scaled fac_data in place: the reference's function would scale it again, so only the transform is redone) */
    fprintf(stderr, "xaacdec_dropin: USAC FD frame refused with FAC state already consumed\n");
    exit(3);
  }
  HIPU(hipMemcpy(u->overlap_data_ptr[ch], d.overlap, (size_t)ccfl * 4, hipMemcpyDeviceToHost));
  HIPU(hipMemcpy(u->output_data_ptr[ch], d.out32, (size_t)ccfl * 4, hipMemcpyDeviceToHost));
  if (td_prev) { /* imdct.c:459-470 / :578-589: the LPD decoder's bass post filter on the float samples, then back to Q15 */
    FLOAT32 *t = u->time_sample_vector[ch];
    WORD32 *o = u->output_data_ptr[ch];
    const FLOAT32 qfac = 1.0f / (FLOAT32)(1 << 15);
    WORD32 err;
    for (k = 0; k < ccfl; k++) t[k] = ((FLOAT32)o[k]) * qfac;
    err = ixheaacd_lpd_bpf_fix(u, seq == EIGHT_SHORT_SEQUENCE, t, u->str_tddec[ch]);
    if (err) return err;
    for (k = 0; k < ccfl; k++) o[k] = (WORD32)(t[k] * (1 << 15));
  }
  dropin_count_usac_imdct(fac_apply, td_prev);
  if (fac_apply && td_prev && !host_fac) dropin_count_usac_fac_on_device();
  return 0;
}
