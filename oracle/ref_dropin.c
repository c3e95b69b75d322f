/*
 * oracle/ref_dropin.c -- TEST INFRASTRUCTURE ONLY.
 *
 * The drop-in, demonstrated: the REAL reference decoder (its own parser, Huffman, TNS, API layer ...) built
 * into oracle/_ref/xaacdec_dropin with -Wl,--wrap=ixheaacd_imdct_process,--wrap=ixheaacd_sbr_dec, so that
 * every call of the two frame-level seams (decoder/ixheaacd_block.h:132, decoder/ixheaacd_sbr_dec.c:662)
 * goes to libxaac_amd's C ABI on the GPU instead of the reference's CPU code: one channel-frame per batch,
 * state converted to the boundary structs, copied to the device, processed, copied back
 * (oracle/ref_convert.h is the mapping; INTEGRATION.md describes the same thing for a batched host).
 * tests/test_dropin_gpu.py decodes whole .aac streams this way and requires the output file to be byte-
 * identical to the unmodified reference decoder's.  Nothing here is linked into the product.
 */
#include <hip/hip_runtime_api.h>

#include "ref_convert.h"
#include "ixheaacd_block.h"
#include "ixheaacd_aac_imdct.h"
#include "ixheaacd_audioobjtypes.h"
#include "ixheaacd_peak_limiter_struct_def.h"
#include "xaac_amd.h"

static xaac_ctx *g_ctx;
static struct {
  int32_t *spec, *overlap, *out32;
  xaac_ics_info *ics;
  xaac_ovl_state *ovl_state;
  int8_t *qadj;
  int16_t *pcm_in, *pcm_out;
  xaac_sbr_header *hdr;
  xaac_sbr_frame *frame;
  xaac_sbr_state *state;
  xaac_ps_frame *psf;
  xaac_ps_state *pss;
  int32_t *status;
  void *ws;
  uint64_t ws_bytes;
  float *core, *time, *time_r; /* Path A: time_sample_buf in / out (left, right) */
  xaac_esbr_ps_state *epss;
  xaac_esbr_side *side;
  xaac_esbr_state *estate;
  xaac_hbe_state *hbe;         /* Path A: the channel's QMF harmonic transposer */
  void *ews;
  xaac_esbr_pvc_side *pvs;     /* USAC channels: the PVC side info and state (xaac_esbr.h) */
  xaac_esbr_pvc_state *pvst;
} g;
static long g_eld_ana_calls, g_eld_syn_calls;
static long g_usac_fac_dev, g_sbr_ds_calls, g_esbr_ds_calls, g_dft_calls, g_dft_ref_calls, g_esbr_dft_calls;
static long g_imdct_calls, g_sbr_calls, g_lim_calls, g_esbr_calls, g_esbr_harm_calls, g_esbr_usac_calls, g_esbr_pvc_calls, g_esbr_83_calls, g_esbr_41_calls, g_sbr_ref_calls, g_usac_imdct_calls, g_usac_imdct_fac, g_usac_imdct_lpd, g_eld_sbr_calls, g_imdct960_calls, g_imdct_ld_calls;
static struct { int32_t *overlap; int16_t *pcm; uint8_t *shape; } gl; /* AAC-LD / ELD: 3 x 512 overlap words, 512 samples, 2 bytes */

static void die(const char *what) {
  fprintf(stderr, "xaacdec_dropin: %s failed\n", what);
  exit(3);
}
#define HIP(x) do { if ((x) != hipSuccess) die(#x); } while (0)

static void report(void) {
  fprintf(stderr, "xaacdec_dropin: %ld imdct_process and %ld sbr_dec calls ran on the GPU\n", g_imdct_calls, g_sbr_calls);
  fprintf(stderr, "xaacdec_dropin: %ld of the sbr_dec calls with the down-sampled synthesis bank\n", g_sbr_ds_calls);
  fprintf(stderr, "xaacdec_dropin: %ld imdct_process calls of 960-line frames and %ld of AAC-LD / ELD frames ran on the GPU\n",
          g_imdct960_calls, g_imdct_ld_calls);
  fprintf(stderr, "xaacdec_dropin: %ld LD / ELD analysis-bank and %ld synthesis-bank calls ran on the GPU\n", g_eld_ana_calls, g_eld_syn_calls);
  fprintf(stderr, "xaacdec_dropin: %ld whole low-delay SBR calls (AAC-ELD: banks + core) ran on the GPU\n", g_eld_sbr_calls);
  fprintf(stderr, "xaacdec_dropin: %ld peak_limiter_process calls ran on the GPU\n", g_lim_calls);
  fprintf(stderr, "xaacdec_dropin: %ld sbr_dec calls took the eSBR (Path A) branch on the GPU\n", g_esbr_calls);
  fprintf(stderr, "xaacdec_dropin: %ld of them with harmonic patching (the QMF transposer's output)\n", g_esbr_harm_calls);
  fprintf(stderr, "xaacdec_dropin: %ld of the eSBR calls with the down-sampled synthesis bank(s)\n", g_esbr_ds_calls);
  fprintf(stderr, "xaacdec_dropin: %ld of them for USAC channels, %ld sbr_dec calls left to the reference\n", g_esbr_usac_calls, g_sbr_ref_calls);
  fprintf(stderr, "xaacdec_dropin: %ld USAC fd_frm_dec calls ran on the GPU, %ld with a FAC signal, %ld behind an LPD frame\n", g_usac_imdct_calls,
          g_usac_imdct_fac, g_usac_imdct_lpd);
  fprintf(stderr, "xaacdec_dropin: %ld FAC signals made on the device (ixheaacd_cal_fac_data)\n", g_usac_fac_dev);
  fprintf(stderr, "xaacdec_dropin: %ld of the USAC calls were PVC frames (PVC decoder + the adjuster's PVC branch on the GPU)\n", g_esbr_pvc_calls);
  fprintf(stderr, "xaacdec_dropin: %ld of the eSBR calls with the DFT transposer inside the chain (-esbr_hq:1)\n", g_esbr_dft_calls);
  fprintf(stderr, "xaacdec_dropin: %ld dft_hbe_apply calls (-esbr_hq:1: the DFT harmonic transposer) ran on the GPU, %ld left to the reference\n", g_dft_calls,
          g_dft_ref_calls);
  fprintf(stderr, "xaacdec_dropin: %ld of the USAC calls at 8:3 SBR (24-channel bank), %ld at 4:1 (16-channel bank, 64 slots)\n", g_esbr_83_calls, g_esbr_41_calls);
}

static void setup(void) {
  uint64_t a, b;
  if (g_ctx) return;
  if (xaac_create(&g_ctx, 0, NULL) != XAAC_OK) die("xaac_create");
  HIP(hipMalloc((void **)&g.spec, 4096));
  HIP(hipMalloc((void **)&g.overlap, 2048));
  HIP(hipMalloc((void **)&g.out32, 4096));
  HIP(hipMalloc((void **)&g.ics, sizeof(xaac_ics_info)));
  HIP(hipMalloc((void **)&g.ovl_state, sizeof(xaac_ovl_state)));
  HIP(hipMalloc((void **)&g.qadj, 4));
  HIP(hipMalloc((void **)&g.pcm_in, 2048));
  HIP(hipMalloc((void **)&g.pcm_out, 8192));
  HIP(hipMalloc((void **)&g.hdr, sizeof(xaac_sbr_header)));
  HIP(hipMalloc((void **)&g.frame, sizeof(xaac_sbr_frame)));
  HIP(hipMalloc((void **)&g.state, sizeof(xaac_sbr_state)));
  HIP(hipMalloc((void **)&g.psf, sizeof(xaac_ps_frame)));
  HIP(hipMalloc((void **)&g.pss, sizeof(xaac_ps_state)));
  HIP(hipMalloc((void **)&g.status, 4));
  a = xaac_sbr_lp_workspace_bytes(1);
  b = xaac_sbr_hq_workspace_bytes(1, 1);
  g.ws_bytes = a > b ? a : b;
  HIP(hipMalloc(&g.ws, g.ws_bytes));
  HIP(hipMalloc((void **)&g.core, 4096));
  HIP(hipMalloc((void **)&g.time, 16384)); /* 4096 floats: a 4:1 frame */
  HIP(hipMalloc((void **)&g.time_r, 8192));
  HIP(hipMalloc((void **)&g.epss, sizeof(xaac_esbr_ps_state)));
  HIP(hipMalloc((void **)&g.side, sizeof(xaac_esbr_side)));
  HIP(hipMalloc((void **)&g.estate, sizeof(xaac_esbr_state)));
  HIP(hipMalloc((void **)&g.pvs, sizeof(xaac_esbr_pvc_side)));
  HIP(hipMalloc((void **)&g.pvst, sizeof(xaac_esbr_pvc_state)));
  HIP(hipMalloc((void **)&g.hbe, sizeof(xaac_hbe_state)));
  HIP(hipMalloc(&g.ews, xaac_esbr_workspace_bytes(1) + xaac_esbr_workspace_bytes_ratio(1, XAAC_ESBR_RATIO_4_1)));
  HIP(hipMalloc((void **)&gl.overlap, 3 * 512 * 4));
  HIP(hipMalloc((void **)&gl.pcm, 512 * 2));
  HIP(hipMalloc((void **)&gl.shape, 2));
  if (!g_sbr_ref_calls) atexit(report);
}

/* for ref_dropin_usac.c (the USAC FD seam lives in a file of its own: ia_usac_data_struct's headers) */
xaac_ctx *dropin_ctx(void) {
  setup();
  return g_ctx;
}
void dropin_count_usac_imdct(int with_fac, int behind_lpd) {
  g_usac_imdct_calls++;
  g_usac_imdct_fac += with_fac != 0;
  g_usac_imdct_lpd += behind_lpd != 0;
}
void dropin_count_usac_fac_on_device(void) { g_usac_fac_dev++; }


/* developer aid: the numeric code behind the command-line tool's "error unlisted" */
int __real_ixheaacd_error_handler(void *info, char *context, int code);
int __wrap_ixheaacd_error_handler(void *info, char *context, int code) {
  if (code && getenv("XAAC_DROPIN_DEBUG")) fprintf(stderr, "xaacdec_dropin: error code 0x%08x\n", (unsigned)code);
  return __real_ixheaacd_error_handler(info, context, code);
}

/* ---- seam 1: ixheaacd_imdct_process (core AAC back-end, 1024-sample frames) ------------------------- */
VOID __real_ixheaacd_imdct_process(ia_aac_dec_overlap_info *, WORD32 *, ia_ics_info_struct *, VOID *, const WORD16,
                                   WORD32 *, ia_aac_dec_tables_struct *, WORD32, WORD32, WORD);

VOID __wrap_ixheaacd_imdct_process(ia_aac_dec_overlap_info *oi, WORD32 *spec, ia_ics_info_struct *ics, VOID *out,
                                   const WORD16 ch_fac, WORD32 *scratch, ia_aac_dec_tables_struct *tabs,
                                   WORD32 object_type, WORD32 ld_mps_present, WORD slot_element) {
  xaac_imdct_batch b;
  xaac_ics_info hi;
  xaac_ovl_state hs;
  static int32_t tmp[1024];
  int8_t q;
  int i;
  const int ld = object_type == AOT_ER_AAC_LD || object_type == AOT_ER_AAC_ELD;
  if (!getenv("XAAC_DROPIN_PASS_IMDCT") && ld && !ld_mps_present && (ics->frame_length == 512 || ics->frame_length == 480) &&
      ics->window_sequence == ONLY_LONG_SEQUENCE && oi->window_sequence == ONLY_LONG_SEQUENCE) {
    /* AAC-LD / AAC-ELD: lpfuncs.c:385-486 through xaac_imdct_ld_process_batch; PCM16 comes back (out_samples - slot_element) */
    const int fl = ics->frame_length, eld = object_type == AOT_ER_AAC_ELD, nov = eld ? 3 * fl : fl / 2;
    static int16_t pcm[512];
    xaac_imdct_ld_batch lb;
    uint8_t sh[2];
    WORD16 *o16 = (WORD16 *)out - slot_element;
    setup();
    sh[0] = (uint8_t)ics->window_shape;
    sh[1] = (uint8_t)oi->window_shape;
    HIP(hipMemcpy(g.spec, spec, 4 * fl, hipMemcpyHostToDevice));
    HIP(hipMemcpy(gl.overlap, oi->ptr_overlap_buf, 4 * nov, hipMemcpyHostToDevice));
    HIP(hipMemcpy(gl.shape, sh, 2, hipMemcpyHostToDevice));
    memset(&lb, 0, sizeof(lb));
    lb.n_ch = 1;
    lb.ch_fac = 1;
    lb.frame_length = fl;
    lb.eld = eld;
    lb.spec = g.spec;
    lb.window_shape = gl.shape;
    lb.overlap = gl.overlap;
    lb.shape_prev = gl.shape + 1;
    lb.pcm16 = gl.pcm;
    if (xaac_imdct_ld_process_batch(g_ctx, &lb) != XAAC_OK || xaac_sync(g_ctx) != XAAC_OK) die("xaac_imdct_ld_process_batch");
    HIP(hipMemcpy(pcm, gl.pcm, 2 * fl, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(oi->ptr_overlap_buf, gl.overlap, 4 * nov, hipMemcpyDeviceToHost));
    for (i = 0; i < fl; i++) o16[i * ch_fac] = pcm[i];
    oi->window_shape = ics->window_shape; /* lpfuncs.c:800-801 */
    oi->window_sequence = ics->window_sequence;
    ics->qshift_adj = -2;
    g_imdct_ld_calls++;
    return;
  }
  if (getenv("XAAC_DROPIN_PASS_IMDCT") || (ics->frame_length != 1024 && ics->frame_length != 960) || ld_mps_present || ld) {
    __real_ixheaacd_imdct_process(oi, spec, ics, out, ch_fac, scratch, tabs, object_type, ld_mps_present, slot_element);
    return;
  }
  setup();
  if (ics->frame_length == 960) { /* the 960-line profile: the same descriptor with 960 / 480 */
    hi.window_sequence = (uint8_t)ics->window_sequence;
    hi.window_shape = (uint8_t)ics->window_shape;
    hs.window_sequence = (uint8_t)oi->window_sequence;
    hs.window_shape = (uint8_t)oi->window_shape;
    HIP(hipMemcpy(g.spec, spec, 3840, hipMemcpyHostToDevice));
    HIP(hipMemcpy(g.overlap, oi->ptr_overlap_buf, 1920, hipMemcpyHostToDevice));
    HIP(hipMemcpy(g.ics, &hi, sizeof(hi), hipMemcpyHostToDevice));
    HIP(hipMemcpy(g.ovl_state, &hs, sizeof(hs), hipMemcpyHostToDevice));
    memset(&b, 0, sizeof(b));
    b.n_ch = 1;
    b.ch_fac = 1;
    b.spec = g.spec;
    b.ics = g.ics;
    b.overlap = g.overlap;
    b.state = g.ovl_state;
    b.out32 = g.out32;
    b.qshift_adj = g.qadj;
    b.pcm_mode = XAAC_PCM_LC;
    if (xaac_imdct960_process_batch(g_ctx, &b) != XAAC_OK || xaac_sync(g_ctx) != XAAC_OK) die("xaac_imdct960_process_batch");
    HIP(hipMemcpy(tmp, g.out32, 3840, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(oi->ptr_overlap_buf, g.overlap, 1920, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(&hs, g.ovl_state, sizeof(hs), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(&q, g.qadj, 1, hipMemcpyDeviceToHost));
    for (i = 0; i < 960; i++) ((WORD32 *)out)[i * ch_fac] = tmp[i];
    oi->window_sequence = hs.window_sequence;
    oi->window_shape = hs.window_shape;
    ics->qshift_adj = q;
    g_imdct960_calls++;
    return;
  }
  hi.window_sequence = (uint8_t)ics->window_sequence;
  hi.window_shape = (uint8_t)ics->window_shape;
  hs.window_sequence = (uint8_t)oi->window_sequence;
  hs.window_shape = (uint8_t)oi->window_shape;
  HIP(hipMemcpy(g.spec, spec, 4096, hipMemcpyHostToDevice));
  HIP(hipMemcpy(g.overlap, oi->ptr_overlap_buf, 2048, hipMemcpyHostToDevice));
  HIP(hipMemcpy(g.ics, &hi, sizeof(hi), hipMemcpyHostToDevice));
  HIP(hipMemcpy(g.ovl_state, &hs, sizeof(hs), hipMemcpyHostToDevice));
  memset(&b, 0, sizeof(b));
  b.n_ch = 1;
  b.ch_fac = 1;
  b.spec = g.spec;
  b.ics = g.ics;
  b.overlap = g.overlap;
  b.state = g.ovl_state;
  b.out32 = g.out32; /* the WORD32 block the reference leaves in its output buffer (lpfuncs.c:347) */
  b.qshift_adj = g.qadj;
  b.pcm_mode = XAAC_PCM_LC;
  if (xaac_imdct_process_batch(g_ctx, &b) != XAAC_OK || xaac_sync(g_ctx) != XAAC_OK) die("xaac_imdct_process_batch");
  HIP(hipMemcpy(tmp, g.out32, 4096, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(oi->ptr_overlap_buf, g.overlap, 2048, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(&hs, g.ovl_state, sizeof(hs), hipMemcpyDeviceToHost));
  HIP(hipMemcpy(&q, g.qadj, 1, hipMemcpyDeviceToHost));
  if (getenv("XAAC_DROPIN_CHECK")) { /* developer aid: the reference's own function beside the GPU's answer */
    static int32_t ovl_gpu[512];
    int bad_out = 0, bad_ovl = 0;
    HIP(hipMemcpy(ovl_gpu, g.overlap, 2048, hipMemcpyDeviceToHost));
    __real_ixheaacd_imdct_process(oi, spec, ics, out, ch_fac, scratch, tabs, object_type, ld_mps_present, slot_element);
    for (i = 0; i < 1024; i++) bad_out += ((WORD32 *)out)[i * ch_fac] != tmp[i];
    for (i = 0; i < 512; i++) bad_ovl += oi->ptr_overlap_buf[i] != ovl_gpu[i];
    if (bad_out || bad_ovl || ics->qshift_adj != q)
      fprintf(stderr, "xaacdec_dropin: imdct call %ld differs: out %d overlap %d q %d/%d (seq %d shape %d prev %d %d ch_fac %d)\n",
              g_imdct_calls, bad_out, bad_ovl, (int)ics->qshift_adj, (int)q, hi.window_sequence, hi.window_shape,
              hs.window_sequence, hs.window_shape, (int)ch_fac);
    g_imdct_calls++;
    return;
  }
  for (i = 0; i < 1024; i++) ((WORD32 *)out)[i * ch_fac] = tmp[i];
  oi->window_sequence = hs.window_sequence;
  oi->window_shape = hs.window_shape;
  ics->qshift_adj = q;
  g_imdct_calls++;
}

/* ---- the LD / ELD complex QMF banks inside ixheaacd_sbr_dec (which stays the reference's for these profiles) ----------
 * ixheaacd_cplx_anal_qmffilt (generic:590) and ixheaacd_cplx_synt_qmffilt (qmf_dec.c:811) with AOT_ER_AAC_LD / _ELD: bank
 * state pointers <-> offsets, rows gathered to [slot][re 64 | im 64] and scattered back. */
VOID __real_ixheaacd_cplx_anal_qmffilt(const WORD16 *, ia_sbr_scale_fact_struct *, WORD32 **, WORD32 **,
                                       ia_sbr_qmf_filter_bank_struct *, ia_qmf_dec_tables_struct *, WORD32, WORD32, WORD);
VOID __wrap_ixheaacd_cplx_anal_qmffilt(const WORD16 *time, ia_sbr_scale_fact_struct *sf, WORD32 **qre, WORD32 **qim,
                                       ia_sbr_qmf_filter_bank_struct *bank, ia_qmf_dec_tables_struct *t, WORD32 ch_fac,
                                       WORD32 low_pow_flag, WORD aot) {
  static int16_t *d_pcm;
  static int32_t *d_qmf, *d_status;
  static xaac_qmf_ana_eld_state *d_st;
  static int32_t rows[16][128];
  static int16_t pcm[512];
  xaac_qmf_ana_eld_state st;
  xaac_qmf_ana_eld_batch b;
  int32_t status;
  const int ns = bank->num_time_slots;
  int i, k;
  if (getenv("XAAC_DROPIN_PASS_ELD_BANKS") || (aot != AOT_ER_AAC_ELD && aot != AOT_ER_AAC_LD) || low_pow_flag ||
      bank->no_channels != 32 || (ns != 16 && ns != 15)) {
    __real_ixheaacd_cplx_anal_qmffilt(time, sf, qre, qim, bank, t, ch_fac, low_pow_flag, aot);
    return;
  }
  setup();
  if (!d_pcm) {
    HIP(hipMalloc((void **)&d_pcm, sizeof(pcm)));
    HIP(hipMalloc((void **)&d_qmf, sizeof(rows)));
    HIP(hipMalloc((void **)&d_st, sizeof(st)));
    HIP(hipMalloc((void **)&d_status, 4));
  }
  /* what the function does to the bank and the scale factors besides the filtering (generic:609-651) */
  bank->filter_pos += t->qmf_c_eld3 - bank->analy_win_coeff;
  bank->analy_win_coeff = t->qmf_c_eld3;
  sf->st_lb_scale = 0;
  sf->lb_scale = -9;
  bank->cos_twiddle = (WORD16 *)t->sbr_sin_cos_twiddle_l32;
  bank->alt_sin_twiddle = (WORD16 *)t->sbr_alt_sin_twiddle_l32;
  bank->t_cos = (WORD16 *)t->ixheaacd_sbr_t_cos_sin_l32_eld;
  memcpy(st.ring, bank->anal_filter_states, sizeof(st.ring));
  st.wr = (int16_t)(bank->core_samples_buffer - bank->anal_filter_states);
  st.f1 = (int16_t)(bank->filter_pos - t->qmf_c_eld3);
  st.f2 = (int16_t)(bank->filter_2 - t->qmf_c_eld3);
  st.fp = (int16_t)(bank->fp1_anal - bank->anal_filter_states);
  for (i = 0; i < 32 * ns; i++) pcm[i] = time[i * ch_fac];
  for (i = 0; i < ns; i++)
    for (k = 0; k < 64; k++) {
      rows[i][k] = qre[i][k];
      rows[i][64 + k] = qim[i][k];
    }
  HIP(hipMemcpy(d_pcm, pcm, 2 * 32 * ns, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_qmf, rows, 512 * ns, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_st, &st, sizeof(st), hipMemcpyHostToDevice));
  memset(&b, 0, sizeof(b));
  b.n_ch = 1;
  b.n_slots = ns;
  b.usb = bank->usb;
  b.slot_stride = 128;
  b.pcm = d_pcm;
  b.state = d_st;
  b.qmf = d_qmf;
  b.status = d_status;
  if (xaac_qmf_analysis_eld_batch(g_ctx, &b) != XAAC_OK || xaac_sync(g_ctx) != XAAC_OK) die("xaac_qmf_analysis_eld_batch");
  HIP(hipMemcpy(&status, d_status, 4, hipMemcpyDeviceToHost));
  if (status) die("xaac_qmf_analysis_eld_batch: bank state outside the ten phases");
  HIP(hipMemcpy(rows, d_qmf, 512 * ns, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(&st, d_st, sizeof(st), hipMemcpyDeviceToHost));
  for (i = 0; i < ns; i++)
    for (k = 0; k < 64; k++) {
      qre[i][k] = rows[i][k];
      qim[i][k] = rows[i][64 + k];
    }
  memcpy(bank->anal_filter_states, st.ring, sizeof(st.ring));
  bank->core_samples_buffer = bank->anal_filter_states + st.wr;
  bank->filter_pos = t->qmf_c_eld3 + st.f1;
  bank->filter_2 = t->qmf_c_eld3 + st.f2;
  bank->fp1_anal = bank->anal_filter_states + st.fp;
  bank->fp2_anal = bank->anal_filter_states + (32 - st.fp);
  g_eld_ana_calls++;
}

VOID __real_ixheaacd_cplx_synt_qmffilt(WORD32 **, WORD32 **, WORD32, WORD32 **, WORD32 **, ia_sbr_scale_fact_struct *, WORD16 *,
                                       ia_sbr_qmf_filter_bank_struct *, ia_ps_dec_struct *, FLAG, FLAG, ia_sbr_tables_struct *,
                                       ixheaacd_misc_tables *, WORD32, FLAG, WORD32 (*)[64], WORD32);
VOID __wrap_ixheaacd_cplx_synt_qmffilt(WORD32 **qre, WORD32 **qim, WORD32 split, WORD32 **ore, WORD32 **oim,
                                       ia_sbr_scale_fact_struct *sf, WORD16 *time_out, ia_sbr_qmf_filter_bank_struct *bank,
                                       ia_ps_dec_struct *ps, FLAG active, FLAG low_pow_flag, ia_sbr_tables_struct *tabs,
                                       ixheaacd_misc_tables *misc, WORD32 ch_fac, FLAG drc_on, WORD32 drc[][64], WORD32 aot) {
  static int32_t *d_qmf, *d_status, *d_scaled;
  static int16_t *d_scale, *d_pcm;
  static xaac_qmf_syn_eld_state *d_st;
  static int32_t rows[16][128];
  static int16_t pcm[1024];
  xaac_qmf_syn_eld_state st;
  xaac_qmf_syn_eld_batch b;
  int16_t scale[4];
  int32_t status;
  const int ns = bank->num_time_slots;
  ia_qmf_dec_tables_struct *t = tabs->qmf_dec_tables_ptr;
  int i, k;
  if (getenv("XAAC_DROPIN_PASS_ELD_BANKS") || (aot != AOT_ER_AAC_ELD && aot != AOT_ER_AAC_LD) || low_pow_flag || active ||
      drc_on || bank->no_channels != 64 || (ns != 16 && ns != 15)) {
    __real_ixheaacd_cplx_synt_qmffilt(qre, qim, split, ore, oim, sf, time_out, bank, ps, active, low_pow_flag, tabs, misc, ch_fac,
                                      drc_on, drc, aot);
    return;
  }
  setup();
  if (!d_qmf) {
    HIP(hipMalloc((void **)&d_qmf, sizeof(rows)));
    HIP(hipMalloc((void **)&d_scaled, sizeof(rows)));
    HIP(hipMalloc((void **)&d_scale, 8));
    HIP(hipMalloc((void **)&d_pcm, sizeof(pcm)));
    HIP(hipMalloc((void **)&d_st, sizeof(st)));
    HIP(hipMalloc((void **)&d_status, 4));
  }
  bank->cos_twiddle = (WORD16 *)t->sbr_sin_cos_twiddle_l64; /* qmf_dec.c:862-875 */
  bank->alt_sin_twiddle = (WORD16 *)t->sbr_alt_sin_twiddle_l64;
  bank->filter_pos_syn += t->qmf_c_eld - bank->p_filter;
  bank->p_filter = t->qmf_c_eld;
  memcpy(st.ring, bank->filter_states, sizeof(st.ring));
  st.drc_offset = (int16_t)bank->ixheaacd_drc_offset;
  st.phase = (int16_t)(bank->filter_pos_syn - t->qmf_c_eld);
  st.fp = (int16_t)(bank->fp1_syn - bank->filter_states);
  st.sixty4 = (int16_t)bank->sixty4;
  scale[0] = sf->lb_scale;
  scale[1] = sf->ov_lb_scale;
  scale[2] = sf->hb_scale;
  scale[3] = sf->st_syn_scale;
  for (i = 0; i < ns; i++)
    for (k = 0; k < 64; k++) {
      rows[i][k] = qre[i][k];
      rows[i][64 + k] = qim[i][k];
    }
  HIP(hipMemcpy(d_qmf, rows, 512 * ns, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_scale, scale, 8, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_st, &st, sizeof(st), hipMemcpyHostToDevice));
  memset(&b, 0, sizeof(b));
  b.n_ch = 1;
  b.n_slots = ns;
  b.lsb = bank->lsb;
  b.usb = bank->usb;
  b.split = split;
  b.slot_stride = 128;
  b.qmf = d_qmf;
  b.scale = d_scale;
  b.state = d_st;
  b.pcm = d_pcm;
  b.status = d_status;
  b.qmf_scaled = d_scaled;
  if (xaac_qmf_synthesis_eld_batch(g_ctx, &b) != XAAC_OK || xaac_sync(g_ctx) != XAAC_OK) die("xaac_qmf_synthesis_eld_batch");
  HIP(hipMemcpy(&status, d_status, 4, hipMemcpyDeviceToHost));
  if (status) die("xaac_qmf_synthesis_eld_batch: bank state outside the ten phases");
  HIP(hipMemcpy(pcm, d_pcm, 2 * 64 * ns, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(&st, d_st, sizeof(st), hipMemcpyDeviceToHost));
  for (i = 0; i < 64 * ns; i++) time_out[i * ch_fac] = pcm[i];
  memcpy(bank->filter_states, st.ring, sizeof(st.ring));
  bank->ixheaacd_drc_offset = st.drc_offset;
  bank->filter_pos_syn = t->qmf_c_eld + st.phase;
  bank->fp1_syn = bank->filter_states + st.fp;
  bank->sixty4 = st.sixty4;
  bank->fp2_syn = bank->fp1_syn + bank->sixty4;
  /* the reference rescales the matrix in place (qmf_dec.c:937-953), hands the rescaled rows on (:966-976) and then uses its input
     rows as work space: the kernel's optional qmf_scaled output serves both */
  HIP(hipMemcpy(rows, d_scaled, 512 * ns, hipMemcpyDeviceToHost));
  for (i = 0; i < ns; i++)
    for (k = 0; k < 64; k++) {
      qre[i][k] = ore[i][k] = rows[i][k];
      qim[i][k] = oim[i][k] = rows[i][64 + k];
    }
  g_eld_syn_calls++;
}

/* ---- seam 2: ixheaacd_sbr_dec (fixed-point Path B: low-power, HQ, HQ + parametric stereo) ------------ */
WORD32 __real_ixheaacd_sbr_dec(ia_sbr_dec_struct *, WORD16 *, ia_sbr_header_data_struct *,
                               ia_sbr_frame_info_data_struct *, ia_sbr_prev_frame_data_struct *, ia_ps_dec_struct *,
                               ia_sbr_qmf_filter_bank_struct *, ia_sbr_scale_fact_struct *, FLAG, FLAG, WORD32 *,
                               ia_sbr_tables_struct *, ixheaacd_misc_tables *, WORD, ia_pvc_data_struct *, FLAG,
                               WORD32[][64], WORD32, WORD32, VOID *, WORD32, WORD32);

/* ---- -esbr_hq:1: the DFT harmonic transposer's struct members <-> include/xaac_hbe.h (xaac_hbe_dft_state / _cfg) ---- */
static struct { xaac_hbe_dft_state *st; xaac_hbe_dft_cfg *cfg; float *coef, *q, *pv; int32_t *par; } gd;
static void dft_alloc(void) {
  if (gd.st) return;
  HIP(hipMalloc((void **)&gd.st, sizeof(xaac_hbe_dft_state)));
  HIP(hipMalloc((void **)&gd.cfg, sizeof(xaac_hbe_dft_cfg)));
  HIP(hipMalloc((void **)&gd.coef, 2 * 64 * 128 * 4));
  HIP(hipMalloc((void **)&gd.q, 2 * 2048 * 4));
  HIP(hipMalloc((void **)&gd.pv, 2 * 34 * 64 * 4));
  HIP(hipMalloc((void **)&gd.par, 16));
}
static int dft_sizes_fit(const ia_esbr_hbe_txposer_struct *t) {
  const int ana0 = t->ana_fft_size[0], syn0 = t->syn_fft_size[0];
  return ana0 >= 0 && ana0 <= XAAC_HBE_DFT_MAX_ANA && syn0 >= 0 && syn0 <= XAAC_HBE_DFT_MAX_SYN && ana0 == 32 * t->synth_size &&
         syn0 == 16 * t->analy_size;
}
/* state and windows up (the windows and the analysis bank's matrices with every call: a test harness, 90 KB a call) */
static void dft_upload(const ia_esbr_hbe_txposer_struct *t) {
  static xaac_hbe_dft_state st;
  static xaac_hbe_dft_cfg cfg;
  const int ana0 = t->ana_fft_size[0], syn0 = t->syn_fft_size[0];
  int tr, o;
  dft_alloc();
  memset(&st, 0, sizeof(st));
  memcpy(st.input_buf, t->ptr_input_buf, sizeof(float) * 2 * ana0);
  memcpy(st.output_buf, t->ptr_output_buf, sizeof(float) * 4 * syn0);
  memcpy(st.synth_buf, t->synth_buf, sizeof(st.synth_buf));
  memcpy(st.anal.analy_buf, t->analy_buf, sizeof(st.anal.analy_buf));
  st.anal.analy_size = t->analy_size;
  st.anal.a_start = t->a_start;
  st.synth_size = t->synth_size;
  st.k_start = t->k_start;
  st.start_band = t->start_band;
  st.end_band = t->end_band;
  st.max_stretch = t->max_stretch;
  for (o = 0; o < 6; o++) st.x_over_qmf[o] = t->x_over_qmf[o];
  memset(&cfg, 0, sizeof(cfg));
  memcpy(cfg.anal_window, t->anal_window, sizeof(float) * ana0);
  memcpy(cfg.synth_window, t->synth_window, sizeof(float) * syn0);
  for (tr = 0; tr < 3; tr++)
    for (o = 0; o < 2; o++) memcpy(cfg.fd_win[tr][o], t->fd_win_buf[tr][o], sizeof(cfg.fd_win[tr][o]));
  HIP(hipMemcpy(gd.st, &st, sizeof(st), hipMemcpyHostToDevice));
  HIP(hipMemcpy(gd.cfg, &cfg, sizeof(cfg), hipMemcpyHostToDevice));
  HIP(hipMemcpy(gd.coef, t->str_dft_hbe_anal_coeff.real, 64 * 128 * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(gd.coef + 64 * 128, t->str_dft_hbe_anal_coeff.imag, 64 * 128 * 4, hipMemcpyHostToDevice));
}
/* the signals and delay lines back; returns last_status */
static int dft_download(ia_esbr_hbe_txposer_struct *t) {
  static xaac_hbe_dft_state st;
  const int ana0 = t->ana_fft_size[0], syn0 = t->syn_fft_size[0];
  HIP(hipMemcpy(&st, gd.st, sizeof(st), hipMemcpyDeviceToHost));
  if (st.last_status) return st.last_status;
  memcpy(t->ptr_input_buf, st.input_buf, sizeof(float) * 2 * ana0);
  memcpy(t->ptr_output_buf, st.output_buf, sizeof(float) * 4 * syn0);
  memcpy(t->synth_buf, st.synth_buf, sizeof(st.synth_buf));
  memcpy(t->analy_buf, st.anal.analy_buf, sizeof(st.anal.analy_buf));
  return 0;
}

WORD32 __wrap_ixheaacd_sbr_dec(ia_sbr_dec_struct *d, WORD16 *time_data, ia_sbr_header_data_struct *h,
                               ia_sbr_frame_info_data_struct *f, ia_sbr_prev_frame_data_struct *p,
                               ia_ps_dec_struct *ps, ia_sbr_qmf_filter_bank_struct *synth_r,
                               ia_sbr_scale_fact_struct *sf_r, FLAG apply, FLAG low_pow, WORD32 *work,
                               ia_sbr_tables_struct *tabs, ixheaacd_misc_tables *common, WORD ch_fac,
                               ia_pvc_data_struct *pvc, FLAG drc_on, WORD32 drc[][64], WORD32 aot, WORD32 ldmps,
                               VOID *self, WORD32 mps, WORD32 ec) {
  static xaac_sbr_header hd;
  static xaac_sbr_frame fr;
  static xaac_sbr_state st;
  static xaac_ps_frame psf;
  static xaac_ps_state pss;
  static int16_t in[1024], outp[4096];
  int32_t status = 0;
  const int with_ps = !low_pow && ps && h->channel_mode == PS_STEREO;
  const int ps_on = with_ps && apply;
  int i, rc;
  /* the reference's default path (-esbr:1) on HE-AAC mono / stereo channels: Path A on the GPU.  (For such streams the
     reference also runs its QMF harmonic transposer every frame -- hbe_flag is forced on, sbrdecoder.c:399-403 -- but
     with sbr_patching_mode 1 nothing reads what it produces: ixheaacd_generate_hf takes the LPP branch; the chain runs it
     too, see below.)  Everything the branch
     at sbr_dec.c:816-1009 does for such a channel -- history shift, analysis, HF generator, envelope adjuster,
     regrouping, synthesis -- is one xaac_esbr_sbr_process_batch call; the state lives in the reference's structs
     between calls (to_esbr_state / from_esbr_state) */
  /* USAC channels (stereoConfigIndex 0, ORIG_SBR frames) come through the same branch: with a transposer like the AAC ones,
     without one with codec_x_delay 0 (sbr_dec.c:819-826; xaac_esbr.h: XAAC_ESBR_USAC / _NO_X_DELAY) */
  /* -esbr_hq:1 (the DFT transposer in the QMF one's place): the same call with xaac_esbr_sbr_batch.hbe_dft_state, where the
     reference has transforms for the transposer's sizes; the reset-time calls of ixheaacd_applysbr go through the seam on
     ixheaacd_dft_hbe_apply at the end of this file */
  if (h->enh_sbr && aot != AOT_ER_AAC_ELD && aot != AOT_ER_AAC_LD &&
      (!h->esbr_hq || (!h->usac_flag && h->hbe_flag && d->p_hbe_txposer != NULL && dft_sizes_fit(d->p_hbe_txposer) && !getenv("XAAC_DROPIN_NO_DFT_CHAIN"))) &&
      (h->usac_flag ? ((!h->hbe_flag || d->p_hbe_txposer != NULL) && f->stereo_config_idx == 0 && !getenv("XAAC_DROPIN_NO_USAC"))
                    : (h->hbe_flag && d->p_hbe_txposer != NULL)) &&
      (h->channel_mode == PS_STEREO ? (ps != NULL && synth_r != NULL && !ps->use_34_st_bands && !ps->use_pca_rot_flg && ps->ps_mode == 0)
                                    : !h->enh_sbr_ps) &&
      !drc_on && !ldmps && !mps && !f->mps_sbr_flag &&
      (f->sbr_mode != PVC_SBR || (h->usac_flag && pvc != NULL && !low_pow && !getenv("XAAC_DROPIN_NO_PVC"))) &&
      /* the three SBR ratios: 2:1; 8:3 (USAC, 768-sample frames, 24-channel bank; the header counts 12 time slots there: core
         frame / 64); 4:1 (USAC, 16-channel bank, 64 slots) without a transposer */
      ((h->num_time_slots == 16 && d->str_codec_qmf_bank.no_channels == 32) ||
       (h->usac_flag && h->sbr_ratio_idx == SBR_UPSAMPLE_IDX_8_3 && h->num_time_slots == 12 && d->str_codec_qmf_bank.no_channels == 24 &&
        !getenv("XAAC_DROPIN_NO_RATIOS")) ||
       (h->usac_flag && h->sbr_ratio_idx == SBR_UPSAMPLE_IDX_4_1 && h->num_time_slots == 16 && d->str_codec_qmf_bank.no_channels == 16 &&
        !h->hbe_flag && !getenv("XAAC_DROPIN_NO_RATIOS"))) &&
      (d->str_synthesis_qmf_bank.no_channels == 64 ||   /* or the down-sampled bank(s): -dsample:1, output rates above 48 kHz */
       (d->str_synthesis_qmf_bank.no_channels == 32 && (h->channel_mode != PS_STEREO || synth_r->no_channels == 32) &&
        !getenv("XAAC_DROPIN_NO_ESBR_DS"))) &&
      !getenv("XAAC_DROPIN_NO_ESBR")) {
    const int esbr_ds = d->str_synthesis_qmf_bank.no_channels == 32;
    const int ratio = d->str_codec_qmf_bank.no_channels == 16 ? XAAC_ESBR_RATIO_4_1
                      : (d->str_codec_qmf_bank.no_channels == 24 ? XAAC_ESBR_RATIO_8_3 : XAAC_ESBR_RATIO_2_1);
    const int out_floats = (ratio == XAAC_ESBR_RATIO_4_1 ? 4096 : 2048) >> esbr_ds;
    static xaac_esbr_side sd;
    static xaac_esbr_state est;
    static xaac_esbr_ps_state epss;
    static xaac_hbe_state hbs;
    static xaac_esbr_pvc_side pvs;
    static xaac_esbr_pvc_state pvst;
    xaac_esbr_sbr_batch b;
    const ia_qmf_dec_tables_struct *q = tabs->qmf_dec_tables_ptr;
    const int eps = h->channel_mode == PS_STEREO;
    setup();
    if (eps) { /* the same re-basing for the right channel's bank */
      synth_r->filter_pos_syn_32 += q->esbr_qmf_c - synth_r->p_filter_32;
      synth_r->p_filter_32 = q->esbr_qmf_c;
    }
    /* the pointer re-basing ixheaacd_esbr_synthesis_filt_block does on entry (sbr_dec.c:578-580) */
    d->str_synthesis_qmf_bank.filter_pos_syn_32 += q->esbr_qmf_c - d->str_synthesis_qmf_bank.p_filter_32;
    d->str_synthesis_qmf_bank.p_filter_32 = q->esbr_qmf_c;
    to_header(h, d, &hd);
    to_frame(f, apply, &fr);
    to_esbr_side(h, f, &sd);
    to_esbr_state(d, h, f, &est);
    static float in_re[32][64], in_im[32][64]; /* the history rows 0..31 the call starts from: see behind the call */
    memcpy(in_re, est.qmf_re, sizeof(in_re));
    memcpy(in_im, est.qmf_im, sizeof(in_im));
    HIP(hipMemcpy(g.hdr, &hd, sizeof(hd), hipMemcpyHostToDevice));
    HIP(hipMemcpy(g.frame, &fr, sizeof(fr), hipMemcpyHostToDevice));
    HIP(hipMemcpy(g.side, &sd, sizeof(sd), hipMemcpyHostToDevice));
    HIP(hipMemcpy(g.estate, &est, sizeof(est), hipMemcpyHostToDevice));
    /* the channel's harmonic transposer: the reference runs it on every processed frame of such a stream
       (sbr_dec.c:882-909) and frames with sbr_patching_mode 0 (ENHSBR payload) take the HF generator's input from it;
       pre-flattening (pre_proc_flag) only acts on LPP patches (sbrdec_lpfuncs.c:1220), so it is moot for those frames */
    if (h->hbe_flag && h->esbr_hq) {
      dft_upload(d->p_hbe_txposer);
    } else if (h->hbe_flag) {
      to_hbe_state(d->p_hbe_txposer, &hbs);
      HIP(hipMemcpy(g.hbe, &hbs, sizeof(hbs), hipMemcpyHostToDevice));
    }
    HIP(hipMemcpy(g.core, d->time_sample_buf, 4096, hipMemcpyHostToDevice));
    if (getenv("XAAC_DROPIN_TRACE_FILE")) { /* developer aid: what every eSBR call is handed (tools compare it with a capture run's) */
      static FILE *ft;
      if (!ft) ft = fopen(getenv("XAAC_DROPIN_TRACE_FILE"), "wb");
      fwrite(&est, sizeof(est), 1, ft), fwrite(&hbs, sizeof(hbs), 1, ft), fwrite(&hd, sizeof(hd), 1, ft), fwrite(&fr, sizeof(fr), 1, ft);
      fwrite(&sd, sizeof(sd), 1, ft), fwrite(d->time_sample_buf, 4, 1024, ft);
      fflush(ft);
    }
    memset(&b, 0, sizeof(b));
    b.n_ch = 1;
    b.core = g.core;
    b.header = g.hdr;
    b.frame = g.frame;
    b.side = g.side;
    b.state = g.estate;
    b.out = g.time;
    if (eps) {
      to_ps_frame(ps, &psf);
      to_esbr_ps_state(ps, synth_r, &epss);
      HIP(hipMemcpy(g.psf, &psf, sizeof(psf), hipMemcpyHostToDevice));
      HIP(hipMemcpy(g.epss, &epss, sizeof(epss), hipMemcpyHostToDevice));
      b.ps_frame = g.psf;
      b.ps_state = g.epss;
      b.out_r = g.time_r;
    }
    b.status = g.status;
    b.workspace = g.ews;
    b.sbr_ratio = ratio;
    b.down_sample = esbr_ds;
    b.workspace_bytes = xaac_esbr_workspace_bytes_ratio(1, ratio);
    b.hbe_state = (h->hbe_flag && !h->esbr_hq) ? g.hbe : NULL;
    if (h->hbe_flag && h->esbr_hq) {
      b.hbe_dft_state = gd.st;
      b.hbe_dft_cfg_tab = gd.cfg;
      b.hbe_dft_coef_re = gd.coef;
      b.hbe_dft_coef_im = gd.coef + 64 * 128;
    }
    if (h->usac_flag) { /* every USAC call carries the PVC side info and state: ORIG_SBR frames leave what a PVC frame behind them reads */
      to_esbr_pvc_side(h, f, pvc, low_pow, &pvs);
      to_esbr_pvc_state(h, f, pvc, &pvst);
      HIP(hipMemcpy(g.pvs, &pvs, sizeof(pvs), hipMemcpyHostToDevice));
      HIP(hipMemcpy(g.pvst, &pvst, sizeof(pvst), hipMemcpyHostToDevice));
      b.pvc_side = g.pvs;
      b.pvc_state = g.pvst;
    }
    if (xaac_esbr_sbr_process_batch(g_ctx, &b) != XAAC_OK || xaac_sync(g_ctx) != XAAC_OK) die("xaac_esbr_sbr_process_batch");
    HIP(hipMemcpy(&status, g.status, 4, hipMemcpyDeviceToHost));
    if (status && getenv("XAAC_DROPIN_DEBUG")) {
      fprintf(stderr, "xaacdec_dropin: eSBR frame refused: apply %d num_env %d noise_env %d borders %d %d %d nsf %d %d nnf %d nmf %d "
              "sb %d %d qsp %d fs %d fm0 %d lo0 %d hi0 %d loN %d hiN %d\n", fr.apply_processing, fr.num_env, fr.num_noise_env,
              fr.border_vec[0], fr.border_vec[1], fr.border_vec[fr.num_env], hd.num_sf_bands[0], hd.num_sf_bands[1], hd.num_nf_bands,
              sd.num_mf_bands, hd.sub_band_start, hd.sub_band_end, sd.qmf_sb_prev, sd.out_sampling_freq, sd.f_master_tbl[0],
              hd.freq_band_tbl_lo[0], hd.freq_band_tbl_hi[0], hd.freq_band_tbl_lo[hd.num_sf_bands[0]], hd.freq_band_tbl_hi[hd.num_sf_bands[1]]);
    }
    if (status) return status;
    HIP(hipMemcpy(&est, g.estate, sizeof(est), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(d->time_sample_buf, g.time, (size_t)out_floats * 4, hipMemcpyDeviceToHost));
    from_esbr_state(&est, d, h, f);
    if (h->hbe_flag) {
      /* What else the reference's buffers hold behind its own call: rows 0..31 of qmf_buf are the history the call started from
         (shifted down at its top, sbr_dec.c:835-845; for a non-USAC stream rows 2..7 with the byte-counted clear of :866-872).  The
         next call never reads them -- it shifts rows 32.. over them -- but ixheaacd_applysbr does when a header resets the SBR
         decoder: it runs the transposer over rows 6..69 of these buffers before the call (sbrdecoder.c:196-226).  (Found by
         tools/sweep_streams.py through the drop-in: a 32 kHz stream whose first SBR header arrives behind nine frames of audio.) */
      int r;
      if (!h->usac_flag && sd.qmf_sb_prev >= 0 && sd.qmf_sb_prev <= 64)
        for (r = 2; r < 8; r++) {
          memset(&in_re[r][sd.qmf_sb_prev], 0, (size_t)(64 - sd.qmf_sb_prev));
          memset(&in_im[r][sd.qmf_sb_prev], 0, (size_t)(64 - sd.qmf_sb_prev));
        }
      memcpy(d->qmf_buf_real[0], in_re, sizeof(in_re));
      memcpy(d->qmf_buf_imag[0], in_im, sizeof(in_im));
    }
    if (apply && h->hbe_flag && h->esbr_hq) {
      dft_download(d->p_hbe_txposer);
      g_esbr_dft_calls++;
    } else if (apply && h->hbe_flag) {
      HIP(hipMemcpy(&hbs, g.hbe, sizeof(hbs), hipMemcpyDeviceToHost));
      from_hbe_state(&hbs, d->p_hbe_txposer, h);
    }
    if (h->usac_flag) {
      HIP(hipMemcpy(&pvst, g.pvst, sizeof(pvst), hipMemcpyDeviceToHost));
      from_esbr_pvc_state(&pvst, apply != 0, h, f, pvc);
      if (f->sbr_mode == PVC_SBR) g_esbr_pvc_calls++;
    }
    if (eps) { /* right channel out, PS state back, and what the second synthesis call leaves in channel 1's frame data */
      HIP(hipMemcpy(&epss, g.epss, sizeof(epss), hipMemcpyDeviceToHost));
      HIP(hipMemcpy(ps->time_sample_buf[1], g.time_r, (size_t)out_floats * 4, hipMemcpyDeviceToHost));
      from_esbr_ps_state(&epss, ps, synth_r);
      ps->use_34_st_bands_prev = ps->use_34_st_bands;
      ((ia_sbr_frame_info_data_struct *)((ia_handle_sbr_dec_inst_struct)self)->frame_buffer[1])->reset_flag = 0;
    }
    /* what the branch leaves behind for the parser and the next call (sbr_dec.c:962-966, :657, :1006) */
    d->band_count = h->pstr_freq_band_data->sub_band_end;
    f->reset_flag = 0;
    f->prev_sbr_mode = f->sbr_mode;
    g_esbr_calls++;
    if (h->usac_flag) g_esbr_usac_calls++;
    if (esbr_ds) g_esbr_ds_calls++;
    if (ratio == XAAC_ESBR_RATIO_8_3) g_esbr_83_calls++;
    if (ratio == XAAC_ESBR_RATIO_4_1) g_esbr_41_calls++;
    if (apply && f->sbr_patching_mode == 0) g_esbr_harm_calls++;
    return 0;
  }
  if (h->enh_sbr && getenv("XAAC_DROPIN_DEBUG")) {
    static int once;
    if (once++ < 6)
      fprintf(stderr, "xaacdec_dropin: eSBR call not taken: aot %d usac %d hbe %d enh_ps %d chmode %d drc %d ldmps %d mps %d mps_sbr %d "
              "sbr_mode %d ratio %d preproc %d slots %d ana %d syn %d patching %d\n", (int)aot, (int)h->usac_flag, (int)h->hbe_flag,
              (int)h->enh_sbr_ps, (int)h->channel_mode, (int)drc_on, (int)ldmps, (int)mps, (int)f->mps_sbr_flag, (int)f->sbr_mode,
              (int)h->sbr_ratio_idx, (int)h->pre_proc_flag, (int)h->num_time_slots, (int)d->str_codec_qmf_bank.no_channels,
              (int)d->str_synthesis_qmf_bank.no_channels, (int)f->sbr_patching_mode);
  }
  /* AAC-ELD channels: the whole call -- LD analysis bank, low-delay SBR core, LD synthesis bank -- as one xaac_sbr_eld_process_batch
     (the two banks' own seams above stay for what this does not take: LD-MPS, DRC inside the bank, the low-power flag) */
  if (aot == AOT_ER_AAC_ELD && !low_pow && !drc_on && !ldmps && !mps && (h->num_time_slots == 16 || h->num_time_slots == 15) &&
      h->time_step == 1 && d->str_codec_qmf_bank.no_channels == 32 && d->str_synthesis_qmf_bank.no_channels == 64 &&
      d->str_hf_generator.pstr_settings->num_columns == h->num_time_slots && !getenv("XAAC_DROPIN_NO_ELD_SBR")) {
    static xaac_sbr_eld_state est;
    static struct { xaac_sbr_eld_state *st; int16_t *in, *out; int32_t *hand; void *ws; } e;
    static int16_t ein[512], eout[1024];
    static int32_t hand[16][128];
    ia_qmf_dec_tables_struct *t = tabs->qmf_dec_tables_ptr;
    const int ns = h->num_time_slots;
    xaac_sbr_eld_batch b;
    int k;
    setup();
    if (!e.st) {
      HIP(hipMalloc((void **)&e.st, sizeof(est)));
      HIP(hipMalloc((void **)&e.in, sizeof(ein)));
      HIP(hipMalloc((void **)&e.out, sizeof(eout)));
      HIP(hipMalloc((void **)&e.hand, sizeof(hand)));
      HIP(hipMalloc(&e.ws, xaac_sbr_eld_workspace_bytes(1)));
    }
    /* what the two bank functions do to their banks besides the filtering (generic:609-651, qmf_dec.c:862-875) */
    d->str_codec_qmf_bank.cos_twiddle = (WORD16 *)t->sbr_sin_cos_twiddle_l32;
    d->str_codec_qmf_bank.alt_sin_twiddle = (WORD16 *)t->sbr_alt_sin_twiddle_l32;
    d->str_codec_qmf_bank.t_cos = (WORD16 *)t->ixheaacd_sbr_t_cos_sin_l32_eld;
    d->str_synthesis_qmf_bank.cos_twiddle = (WORD16 *)t->sbr_sin_cos_twiddle_l64;
    d->str_synthesis_qmf_bank.alt_sin_twiddle = (WORD16 *)t->sbr_alt_sin_twiddle_l64;
    to_header(h, d, &hd);
    to_frame(f, apply, &fr);
    to_eld_state(d, p, t, &est);
    for (i = 0; i < 32 * ns; i++) ein[i] = time_data[i * ch_fac];
    HIP(hipMemcpy(g.hdr, &hd, sizeof(hd), hipMemcpyHostToDevice));
    HIP(hipMemcpy(g.frame, &fr, sizeof(fr), hipMemcpyHostToDevice));
    HIP(hipMemcpy(e.st, &est, sizeof(est), hipMemcpyHostToDevice));
    HIP(hipMemcpy(e.in, ein, 2 * 32 * ns, hipMemcpyHostToDevice));
    memset(&b, 0, sizeof(b));
    b.n_ch = 1;
    b.n_slots = ns;
    b.in_ch_fac = b.out_ch_fac = 1;
    b.pcm_in = e.in;
    b.header = g.hdr;
    b.frame = g.frame;
    b.state = e.st;
    b.pcm_out = e.out;
    b.status = g.status;
    b.workspace = e.ws;
    b.workspace_bytes = xaac_sbr_eld_workspace_bytes(1);
    b.qmf_handed_on = e.hand;
    if (xaac_sbr_eld_process_batch(g_ctx, &b) != XAAC_OK || xaac_sync(g_ctx) != XAAC_OK) die("xaac_sbr_eld_process_batch");
    HIP(hipMemcpy(&status, g.status, 4, hipMemcpyDeviceToHost));
    if (status) return status;
    HIP(hipMemcpy(&est, e.st, sizeof(est), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(eout, e.out, 2 * 64 * ns, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(hand, e.hand, 512 * ns, hipMemcpyDeviceToHost));
    from_eld_state(&est, d, p, t);
    for (i = 0; i < 64 * ns; i++) time_data[i * ch_fac] = eout[i];
    for (i = 0; i < ns; i++) /* the rescaled rows the synthesis bank hands on (qmf_dec.c:966-976) */
      for (k = 0; k < 64; k++) {
        d->p_arr_qmf_buf_real[i][k] = hand[i][k];
        d->p_arr_qmf_buf_imag[i][k] = hand[i][64 + k];
      }
    g_eld_sbr_calls++;
    return 0;
  }
  /* outside the paths this library covers (USAC / PS / HBE eSBR, LD/ELD, DRC inside the bank, MPS): the reference's own code */
  /* the down-sampled synthesis bank (-dsample:1, or an output rate above 48 kHz: 32 channels, 1024 samples out; sbrdec_initfuncs.c:622)
     is the library's too (down_sample in both descriptors) -- not together with PS, where the reference itself hands the right
     bank half a slot (qmf_dec.c:1117-1119) */
  const int ds = d->str_synthesis_qmf_bank.no_channels == 32;
  if (h->enh_sbr || aot == AOT_ER_AAC_ELD || aot == AOT_ER_AAC_LD || drc_on || ldmps || mps ||
      h->num_time_slots * h->time_step != 32 || (ds && with_ps) || (!ds && d->str_synthesis_qmf_bank.no_channels != 64)) {
    if (!g_ctx && !g_sbr_ref_calls) atexit(report); /* (a stream none of whose calls reaches the library still gets its summary) */
    g_sbr_ref_calls++;
    rc = __real_ixheaacd_sbr_dec(d, time_data, h, f, p, ps, synth_r, sf_r, apply, low_pow, work, tabs, common, ch_fac,
                                 pvc, drc_on, drc, aot, ldmps, self, mps, ec);
    if (rc && getenv("XAAC_DROPIN_DEBUG")) fprintf(stderr, "xaacdec_dropin: the reference's own ixheaacd_sbr_dec returned %d\n", rc);
    return rc;
  }
  setup();
  to_header(h, d, &hd);
  to_frame(f, apply, &fr);
  to_state(d, p, low_pow, &st);
  for (i = 0; i < 1024; i++) in[i] = time_data[i * ch_fac];
  HIP(hipMemcpy(g.hdr, &hd, sizeof(hd), hipMemcpyHostToDevice));
  HIP(hipMemcpy(g.frame, &fr, sizeof(fr), hipMemcpyHostToDevice));
  HIP(hipMemcpy(g.state, &st, sizeof(st), hipMemcpyHostToDevice));
  HIP(hipMemcpy(g.pcm_in, in, sizeof(in), hipMemcpyHostToDevice));
  if (low_pow) {
    xaac_sbr_lp_batch b;
    memset(&b, 0, sizeof(b));
    b.n_ch = 1;
    b.in_ch_fac = b.out_ch_fac = 1;
    b.down_sample = ds;
    b.pcm_in = g.pcm_in;
    b.header = g.hdr;
    b.frame = g.frame;
    b.state = g.state;
    b.pcm_out = g.pcm_out;
    b.status = g.status;
    b.workspace = g.ws;
    b.workspace_bytes = g.ws_bytes;
    rc = xaac_sbr_lp_process_batch(g_ctx, &b);
  } else {
    xaac_sbr_hq_batch b;
    memset(&b, 0, sizeof(b));
    b.n_ch = 1;
    b.in_ch_fac = b.out_ch_fac = 1;
    b.down_sample = ds;
    b.pcm_in = g.pcm_in;
    b.header = g.hdr;
    b.frame = g.frame;
    b.state = g.state;
    if (with_ps) {
      to_ps_frame(ps, &psf);
      to_ps_state(ps, synth_r, sf_r, &pss);
      HIP(hipMemcpy(g.psf, &psf, sizeof(psf), hipMemcpyHostToDevice));
      HIP(hipMemcpy(g.pss, &pss, sizeof(pss), hipMemcpyHostToDevice));
      b.ps_frame = g.psf;
      b.ps_state = g.pss;
    }
    b.pcm_out = g.pcm_out;
    b.status = g.status;
    b.workspace = g.ws;
    b.workspace_bytes = g.ws_bytes;
    rc = xaac_sbr_hq_process_batch(g_ctx, &b);
  }
  if (rc != XAAC_OK || xaac_sync(g_ctx) != XAAC_OK) die("xaac_sbr_*_process_batch");
  HIP(hipMemcpy(&status, g.status, 4, hipMemcpyDeviceToHost));
  if (status) return status; /* the reference returns before touching anything */
  HIP(hipMemcpy(&st, g.state, sizeof(st), hipMemcpyDeviceToHost));
  HIP(hipMemcpy(outp, g.pcm_out, with_ps ? 8192 : 4096, hipMemcpyDeviceToHost));
  from_state(&st, low_pow, d, p);
  if (with_ps) {
    HIP(hipMemcpy(&pss, g.pss, sizeof(pss), hipMemcpyDeviceToHost));
    from_ps_state(&pss, ps, synth_r, sf_r);
    for (i = 0; i < 2048; i++) {
      time_data[i * ch_fac] = outp[2 * i];
      if (ps_on) time_data[i * ch_fac + 1] = outp[2 * i + 1];
    }
  } else {
    for (i = 0; i < (ds ? 1024 : 2048); i++) time_data[i * ch_fac] = outp[i];
  }
  g_sbr_calls++;
  if (ds) g_sbr_ds_calls++;
  return 0;
}

/* ---- seam 3: ixheaacd_peak_limiter_process (AAC-LC post stage, decoder/ixheaacd_api.c:3667) ----------- */
void ref_limiter_to_ref(ia_peak_limiter_struct *r, const xaac_limiter_state *s);
void ref_limiter_from_ref(xaac_limiter_state *s, const ia_peak_limiter_struct *r);
VOID __real_ixheaacd_peak_limiter_process(ia_peak_limiter_struct *, VOID *, UWORD32, UWORD8 *);

VOID __wrap_ixheaacd_peak_limiter_process(ia_peak_limiter_struct *lim, VOID *samples, UWORD32 frame_len,
                                          UWORD8 *qshift_adj) {
  static xaac_limiter_state st;
  static struct { int32_t *x; int8_t *q; xaac_limiter_state *st; void *ws; uint64_t ws_bytes; } d;
  const UWORD32 nch = lim->num_channels;
  xaac_limiter_batch b;
  if (nch < 1 || nch > XAAC_LIM_MAX_CH || lim->attack_time_samples < 1 || lim->attack_time_samples > XAAC_LIM_MAX_ATTACK ||
      frame_len < 1 || frame_len > 1024) { /* outside the boundary struct: stays on the CPU */
    __real_ixheaacd_peak_limiter_process(lim, samples, frame_len, qshift_adj);
    return;
  }
  setup();
  if (!d.x) {
    HIP(hipMalloc((void **)&d.x, 1024 * XAAC_LIM_MAX_CH * 4));
    HIP(hipMalloc((void **)&d.q, 16));
    HIP(hipMalloc((void **)&d.st, sizeof(xaac_limiter_state)));
    d.ws_bytes = xaac_peak_limiter_workspace_bytes(1);
    HIP(hipMalloc(&d.ws, d.ws_bytes));
  }
  memset(&st, 0, sizeof(st));
  ref_limiter_from_ref(&st, lim);
  HIP(hipMemcpy(d.x, samples, (size_t)frame_len * nch * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d.q, qshift_adj, nch, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d.st, &st, sizeof(st), hipMemcpyHostToDevice));
  memset(&b, 0, sizeof(b));
  b.n_streams = 1; b.frame_len = (int32_t)frame_len; b.num_channels = (int32_t)nch;
  b.samples = d.x; b.stride = (int64_t)frame_len * nch; b.qshift_adj = d.q; b.state = d.st;
  b.workspace = d.ws; b.workspace_bytes = d.ws_bytes;
  if (xaac_peak_limiter_process_batch(g_ctx, &b) != XAAC_OK || xaac_sync(g_ctx) != XAAC_OK) die("xaac_peak_limiter_process_batch");
  HIP(hipMemcpy(samples, d.x, (size_t)frame_len * nch * 4, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(&st, d.st, sizeof(st), hipMemcpyDeviceToHost));
  { /* the reference keeps its buffer pointers; only the contents come back */
    FLOAT32 *max_buf = lim->max_buf, *delayed = lim->delayed_input;
    const UWORD32 a = lim->attack_time_samples;
    lim->gain_modified = st.gain_modified;
    lim->pre_smoothed_gain = st.pre_smoothed_gain;
    lim->delayed_input_index = st.delayed_input_index;
    lim->min_gain = st.min_gain;
    lim->max_idx = st.max_idx;
    lim->cir_buf_pnt = st.cir_buf_pnt;
    memcpy(max_buf, st.max_buf, a * sizeof(FLOAT32));
    memcpy(delayed, st.delayed_input, (size_t)a * nch * sizeof(FLOAT32));
  }
  g_lim_calls++;
}


/* ---- -esbr_hq:1: the DFT harmonic transposer (ixheaacd_dft_hbe_apply, hbe_dft_trans.c:771; call sites sbr_dec.c:884 and the
   reset-time one in sbrdecoder.c) -> xaac_hbe_dft_apply_batch_run.  The rest of such a frame's sbr_dec call stays the
   reference's (the Path A hook above leaves esbr_hq frames alone).  The transposer's signals and delay lines live in the
   reference's struct between calls; the windows and the analysis bank's matrices its re-initialisation made go up with every
   call (a test harness: 90 KB a call). */
WORD32 __real_ixheaacd_dft_hbe_apply(ia_esbr_hbe_txposer_struct *t, FLOAT32 qre[][64], FLOAT32 qim[][64], WORD32 num_columns, FLOAT32 pvr[][64],
                                     FLOAT32 pvi[][64], WORD32 pitch_in_bins, FLOAT32 *scratch);
WORD32 __wrap_ixheaacd_dft_hbe_apply(ia_esbr_hbe_txposer_struct *t, FLOAT32 qre[][64], FLOAT32 qim[][64], WORD32 num_columns, FLOAT32 pvr[][64],
                                     FLOAT32 pvi[][64], WORD32 pitch_in_bins, FLOAT32 *scratch) {
  xaac_hbe_dft_apply_batch b;
  int32_t par[3];
  if (num_columns != 32 || !dft_sizes_fit(t) || getenv("XAAC_DROPIN_NO_DFT")) {
    g_dft_ref_calls++;
    return __real_ixheaacd_dft_hbe_apply(t, qre, qim, num_columns, pvr, pvi, pitch_in_bins, scratch);
  }
  setup();
  dft_upload(t);
  par[0] = pitch_in_bins;
  par[1] = t->oversampling_flag ? 1 : 0;
  par[2] = 0;
  HIP(hipMemcpy(gd.q, qre, 2048 * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(gd.q + 2048, qim, 2048 * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(gd.pv, pvr, 34 * 64 * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(gd.pv + 34 * 64, pvi, 34 * 64 * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(gd.par, par, sizeof(par), hipMemcpyHostToDevice));
  memset(&b, 0, sizeof(b));
  b.n_ch = 1;
  b.qmf_re = gd.q;
  b.qmf_im = gd.q + 2048;
  b.pitch_in_bins = gd.par;
  b.oversampling = gd.par + 1;
  b.cfg_tab = gd.cfg;
  b.coef_re = gd.coef;
  b.coef_im = gd.coef + 64 * 128;
  b.state = gd.st;
  b.pv_re = gd.pv;
  b.pv_im = gd.pv + 34 * 64;
  b.status = gd.par + 2;
  if (xaac_hbe_dft_apply_batch_run(g_ctx, &b) != XAAC_OK) die("xaac_hbe_dft_apply_batch_run");
  HIP(hipDeviceSynchronize());
  if (dft_download(t)) { /* sizes without a transform: the reference fails the frame itself */
    g_dft_ref_calls++;
    return __real_ixheaacd_dft_hbe_apply(t, qre, qim, num_columns, pvr, pvi, pitch_in_bins, scratch);
  }
  HIP(hipMemcpy(pvr, gd.pv, 34 * 64 * 4, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(pvi, gd.pv + 34 * 64, 34 * 64 * 4, hipMemcpyDeviceToHost));
  g_dft_calls++;
  return 0;
}
