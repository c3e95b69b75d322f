/*
 * oracle/ref_capture.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Link-time interposer for the real reference decoder: built into oracle/_ref/xaacdec_capture
 * with -Wl,--wrap=ixheaacd_sbr_dec (see Makefile.ref).  Every call of ixheaacd_sbr_dec
 * (decoder/ixheaacd_sbr_dec.c:662) made while decoding a stream is recorded -- inputs,
 * persistent state before and after, output PCM -- in the plain-C formats of
 * include/xaac_sbr.h, to the file named by $XAAC_CAPTURE_FILE.  This file is at the same time
 * the worked example of the reference-side adapter (reference structs -> boundary structs)
 * that INTEGRATION.md describes.  No reference code is copied: it only reads the reference's
 * own structs through the reference's own headers.
 */
#include "ref_convert.h"


static FILE *g_out;
static int g_calls;

WORD32 __real_ixheaacd_sbr_dec(ia_sbr_dec_struct *, WORD16 *, ia_sbr_header_data_struct *,
                               ia_sbr_frame_info_data_struct *, ia_sbr_prev_frame_data_struct *, ia_ps_dec_struct *,
                               ia_sbr_qmf_filter_bank_struct *, ia_sbr_scale_fact_struct *, FLAG, FLAG, WORD32 *,
                               ia_sbr_tables_struct *, ixheaacd_misc_tables *, WORD, ia_pvc_data_struct *, FLAG,
                               WORD32[][64], WORD32, WORD32, VOID *, WORD32, WORD32);

WORD32 __wrap_ixheaacd_sbr_dec(ia_sbr_dec_struct *d, WORD16 *time_data, ia_sbr_header_data_struct *h,
                               ia_sbr_frame_info_data_struct *f, ia_sbr_prev_frame_data_struct *p,
                               ia_ps_dec_struct *ps, ia_sbr_qmf_filter_bank_struct *synth_r,
                               ia_sbr_scale_fact_struct *sf_r, FLAG apply, FLAG low_pow, WORD32 *work,
                               ia_sbr_tables_struct *tabs, ixheaacd_misc_tables *common, WORD ch_fac,
                               ia_pvc_data_struct *pvc, FLAG drc_on, WORD32 drc[][64], WORD32 aot, WORD32 ldmps,
                               VOID *self, WORD32 mps, WORD32 ec) {
  static xaac_sbr_header hd;
  static xaac_sbr_frame fr;
  static xaac_sbr_state st0, st1;
  static xaac_ps_frame psf;
  static xaac_ps_state ps0, ps1;
  int16_t in[1024], outp[2][2048];
  int32_t meta[8];
  WORD32 ret;
  int i, ps_on = (apply && h->channel_mode == PS_STEREO);
  if (!g_out) {
    const char *path = getenv("XAAC_CAPTURE_FILE");
    g_out = fopen(path ? path : "/tmp/xaac_capture.bin", "wb");
  }
  to_header(h, d, &hd);
  to_frame(f, apply, &fr);
  to_state(d, p, low_pow, &st0);
  if (ps_on) {
    to_ps_frame(ps, &psf);
    to_ps_state(ps, synth_r, sf_r, &ps0);
  }
  for (i = 0; i < 1024; i++) in[i] = time_data[i * ch_fac];
  ret = __real_ixheaacd_sbr_dec(d, time_data, h, f, p, ps, synth_r, sf_r, apply, low_pow, work, tabs, common, ch_fac,
                                pvc, drc_on, drc, aot, ldmps, self, mps, ec);
  to_state(d, p, low_pow, &st1);
  if (ps_on) to_ps_state(ps, synth_r, sf_r, &ps1);
  for (i = 0; i < 2048; i++) {
    outp[0][i] = time_data[i * ch_fac];
    outp[1][i] = ps_on ? time_data[i * ch_fac + 1] : 0;
  }
  meta[0] = 0x58414331; /* "XAC1" */
  meta[1] = g_calls++;
  meta[2] = low_pow;
  meta[3] = ch_fac;
  meta[4] = aot;
  meta[5] = ps_on;
  meta[6] = ret;
  meta[7] = h->enh_sbr;
  fwrite(meta, sizeof(meta), 1, g_out);
  fwrite(&hd, sizeof(hd), 1, g_out);
  fwrite(&fr, sizeof(fr), 1, g_out);
  fwrite(&st0, sizeof(st0), 1, g_out);
  fwrite(in, sizeof(in), 1, g_out);
  fwrite(&st1, sizeof(st1), 1, g_out);
  fwrite(outp, sizeof(outp), 1, g_out);
  if (ps_on) { /* HE-AACv2 records carry the PS side info and state as well */
    fwrite(&psf, sizeof(psf), 1, g_out);
    fwrite(&ps0, sizeof(ps0), 1, g_out);
    fwrite(&ps1, sizeof(ps1), 1, g_out);
  }
  fflush(g_out);
  return ret;
}

/* Debug aid: with $XAAC_PS_DUMP set, every ixheaacd_apply_ps call (thumb_ps_dec.c:69) appends
   {slot, left re[64] im[64], right re[64] im[64]} after the call to that file. */
VOID __real_ixheaacd_apply_ps(ia_ps_dec_struct *, WORD32 **, WORD32 **, WORD32 *, WORD32 *, ia_sbr_scale_fact_struct *,
                              WORD16, ia_sbr_tables_struct *, WORD);
VOID __wrap_ixheaacd_apply_ps(ia_ps_dec_struct *ps, WORD32 **lr, WORD32 **li, WORD32 *rr, WORD32 *ri,
                              ia_sbr_scale_fact_struct *sf, WORD16 slot, ia_sbr_tables_struct *t, WORD no_col) {
  static FILE *f;
  const char *path = getenv("XAAC_PS_DUMP");
  if (path && getenv("XAAC_PS_DUMP_PRE")) {
    int32_t s = -1 - slot;
    if (!f) f = fopen(path, "wb");
    fwrite(&s, 4, 1, f);
    fwrite(lr[0], 4, 64, f);
    fwrite(li[0], 4, 64, f);
    fwrite(rr, 4, 64, f);
    fwrite(ri, 4, 64, f);
  }
  __real_ixheaacd_apply_ps(ps, lr, li, rr, ri, sf, slot, t, no_col);
  if (path) {
    int32_t s = slot;
    if (!f) f = fopen(path, "wb");
    fwrite(&s, 4, 1, f);
    fwrite(lr[0], 4, 64, f);
    fwrite(li[0], 4, 64, f);
    fwrite(rr, 4, 64, f);
    fwrite(ri, 4, 64, f);
    fflush(f);
  }
}

/* Debug aid: with $XAAC_SYN_DUMP set, every ixheaacd_cplx_synt_qmffilt call (qmf_dec.c:811) appends
   {active, lb, ov_lb, hb, ps, st_syn scales, lsb, usb, 32 rows x (64 re | 64 im)} on entry. */
VOID __real_ixheaacd_cplx_synt_qmffilt(WORD32 **, WORD32 **, WORD32, WORD32 **, WORD32 **, ia_sbr_scale_fact_struct *,
                                       WORD16 *, ia_sbr_qmf_filter_bank_struct *, ia_ps_dec_struct *, FLAG, FLAG,
                                       ia_sbr_tables_struct *, ixheaacd_misc_tables *, WORD32, FLAG, WORD32[][64],
                                       WORD32);
VOID __wrap_ixheaacd_cplx_synt_qmffilt(WORD32 **re, WORD32 **im, WORD32 split, WORD32 **ore, WORD32 **oim,
                                       ia_sbr_scale_fact_struct *sf, WORD16 *out, ia_sbr_qmf_filter_bank_struct *bank,
                                       ia_ps_dec_struct *ps, FLAG active, FLAG low_pow, ia_sbr_tables_struct *t,
                                       ixheaacd_misc_tables *c, WORD32 ch_fac, FLAG drc_on, WORD32 drc[][64],
                                       WORD32 aot) {
  static FILE *f;
  const char *path = getenv("XAAC_SYN_DUMP");
  if (path && !low_pow) {
    int32_t m[8] = {active, sf->lb_scale, sf->ov_lb_scale, sf->hb_scale, sf->ps_scale, sf->st_syn_scale, bank->lsb, bank->usb};
    int i;
    if (!f) f = fopen(path, "wb");
    fwrite(m, 4, 8, f);
    for (i = 0; i < 32; i++) {
      fwrite(re[i], 4, 64, f);
      fwrite(im[i], 4, 64, f);
    }
    fflush(f);
  }
  __real_ixheaacd_cplx_synt_qmffilt(re, im, split, ore, oim, sf, out, bank, ps, active, low_pow, t, c, ch_fac, drc_on, drc,
                                    aot);
}
