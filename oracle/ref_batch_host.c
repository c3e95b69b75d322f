/*
 * oracle/ref_batch_host.c -- the batched multi-stream host of INTEGRATION.md, built around the REAL reference
 * decoder (test infrastructure + integration demonstrator; nothing here is linked into the product library).
 *
 *   xaacdec_batch [-esbr:0 ...] -- <group> [<group> ...]      group = N:input.aac:output-prefix
 *
 * Every group is N independent decoder instances of one stream (instance i writes <output-prefix>.<i>.wav).  An
 * instance is the reference's own command-line decoder, unmodified -- parser, Huffman, TNS, API layer, file I/O --
 * running as a forked child process (the test bench keeps its state in globals, so instances cannot share a process).
 * Its three frame-level seams (ixheaacd_imdct_process, ixheaacd_sbr_dec, ixheaacd_peak_limiter_process) are diverted
 * at link time: the child converts the call's operands to the boundary structs of include/xaac_sbr.h DIRECTLY INTO ITS
 * ROW of the group's staging arrays -- one shared, page-locked (hipHostRegister) region per group, field-major, i.e.
 * exactly the arrays xaac_*_process_batch takes -- and sleeps.  When every live instance of a group has arrived (they
 * decode the same stream type, so they arrive at the same seam), the parent -- the only process that touches HIP --
 * issues on the group's own HIP stream: hipMemcpyAsync of the operand arrays, ONE xaac_*_process_batch for the whole
 * group, hipMemcpyAsync of the results back into the staging arrays, an event.  It does not wait: it goes on serving
 * the other groups, whose children are parsing their next access unit meanwhile (copies, kernels and CPU parsing of
 * different groups overlap -- the double buffering of INTEGRATION.md §4, with as many buffers as there are groups).
 * When a group's event has fired the children are woken and copy their results back into the reference's structs.
 *
 * Per-stream state stays where the reference keeps it (its own structs on the host) and crosses PCIe in both
 * directions on every call, like in xaacdec_dropin: this host measures the path INCLUDING parse, conversion and
 * PCIe.  (A host that owns the decoder instances can leave the state on the GPU between frames -- that is what the
 * C ABI is laid out for and what bench.py measures.)
 *
 * The parent prints one JSON line: streams, seam calls, wall time.  tests/test_batch_host_gpu.py requires every
 * instance's output file to be byte-identical to the unmodified reference decoder's.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <semaphore.h>
#include <signal.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

#include "ref_convert.h"
#include "ixheaacd_block.h"
#include "ixheaacd_aac_imdct.h"
#include "ixheaacd_audioobjtypes.h"
#include "ixheaacd_peak_limiter_struct_def.h"
#include "xaac_amd.h"

int ref_cli_main(int argc, char **argv); /* the reference's test/decoder/ixheaacd_main.c: main, renamed at compile time */
void ref_limiter_from_ref(xaac_limiter_state *s, const ia_peak_limiter_struct *r);

enum { REQ_NONE = 0, REQ_IMDCT, REQ_SBR_LP, REQ_SBR_HQ, REQ_SBR_PS, REQ_ESBR, REQ_ESBR_PS, REQ_LIM, REQ_KINDS };
#define LIM_MAX_CH 2 /* the streams this host is run on are mono / stereo (wider ones stay on the CPU path) */

/* per-instance control word + semaphore */
typedef struct {
  sem_t done;
  volatile int req;   /* REQ_* while the instance waits */
  volatile int lim_nch, lim_len;
  volatile int dft;   /* the eSBR request in flight carries a DFT transposer (-esbr_hq:1) */
  long calls[REQ_KINDS];
} inst_t;

/* one group's staging arrays (all in one shared mapping); n = instances */
typedef struct {
  int n;
  atomic_int pending, live;
  inst_t *inst;
  /* IMDCT */
  int32_t *spec, *ovl, *out32;
  xaac_ics_info *ics;
  xaac_ovl_state *ost;
  int8_t *qadj;
  /* SBR */
  xaac_sbr_header *hdr;
  xaac_sbr_frame *frm;
  xaac_sbr_state *sst;
  xaac_ps_frame *psf;
  xaac_ps_state *pss;
  int16_t *pin, *pout;
  int32_t *status;
  /* SBR, the reference's default float path (eSBR) */
  xaac_esbr_side *esd;
  xaac_esbr_state *est;
  xaac_esbr_ps_state *eps;
  xaac_hbe_state *ehb; /* the channels' QMF harmonic transposers (run on every processed frame, as the reference does) */
  /* -esbr_hq:1: the DFT transposers in their place: state, the windows and the analysis bank's matrices the reference's
     re-initialisation made (one configuration per instance) */
  xaac_hbe_dft_state *edf;
  xaac_hbe_dft_cfg *ecf;
  float *ecoef; /* [2][n][64][128]: the real matrices of all instances, then the imaginary ones */
  int32_t *ecfg; /* [n]: i */
  float *ecore, *etime, *etime_r;
  /* limiter */
  int32_t *lx;
  int8_t *lq;
  xaac_limiter_state *lst;
} group_t;

typedef struct {
  sem_t wake;
  int n_groups;
} shared_hdr_t;

static shared_hdr_t *g_hdr;
static group_t *g_grp;      /* [n_groups], in shared memory (pointers are valid in every process: mapped before fork) */
static group_t *my_grp;     /* child: its group */
static int my_idx = -1;     /* child: its row */

static void *carve(char **cur, size_t bytes) {
  char *r = *cur;
  *cur += (bytes + 255) & ~(size_t)255;
  return r;
}

static size_t group_bytes(int n) {
  size_t per = 4096 + 2048 + 4096 + 8 + 8 + 8 + sizeof(xaac_sbr_header) + sizeof(xaac_sbr_frame) + sizeof(xaac_sbr_state) +
               sizeof(xaac_ps_frame) + sizeof(xaac_ps_state) + 2048 + 8192 + 4 + 1024 * LIM_MAX_CH * 4 + 8 +
               sizeof(xaac_limiter_state) + sizeof(inst_t) + sizeof(xaac_esbr_side) + sizeof(xaac_esbr_state) +
               sizeof(xaac_esbr_ps_state) + sizeof(xaac_hbe_state) + 4096 + 8192 + 8192 + sizeof(xaac_hbe_dft_state) +
               sizeof(xaac_hbe_dft_cfg) + 2 * 64 * 128 * 4 + 4;
  return (size_t)n * per + 128 * 256;
}

static void group_layout(group_t *g, int n, char *base) {
  char *cur = base;
  g->n = n;
  g->inst = carve(&cur, sizeof(inst_t) * n);
  g->spec = carve(&cur, (size_t)n * 4096);
  g->ovl = carve(&cur, (size_t)n * 2048);
  g->out32 = carve(&cur, (size_t)n * 4096);
  g->ics = carve(&cur, sizeof(xaac_ics_info) * n);
  g->ost = carve(&cur, sizeof(xaac_ovl_state) * n);
  g->qadj = carve(&cur, n);
  g->hdr = carve(&cur, sizeof(xaac_sbr_header) * n);
  g->frm = carve(&cur, sizeof(xaac_sbr_frame) * n);
  g->sst = carve(&cur, sizeof(xaac_sbr_state) * n);
  g->psf = carve(&cur, sizeof(xaac_ps_frame) * n);
  g->pss = carve(&cur, sizeof(xaac_ps_state) * n);
  g->pin = carve(&cur, (size_t)n * 2048);
  g->pout = carve(&cur, (size_t)n * 8192);
  g->status = carve(&cur, (size_t)n * 4);
  g->lx = carve(&cur, (size_t)n * 1024 * LIM_MAX_CH * 4);
  g->lq = carve(&cur, (size_t)n * LIM_MAX_CH);
  g->lst = carve(&cur, sizeof(xaac_limiter_state) * n);
  g->esd = carve(&cur, sizeof(xaac_esbr_side) * n);
  g->est = carve(&cur, sizeof(xaac_esbr_state) * n);
  g->eps = carve(&cur, sizeof(xaac_esbr_ps_state) * n);
  g->ehb = carve(&cur, sizeof(xaac_hbe_state) * n);
  g->ecore = carve(&cur, (size_t)n * 4096);
  g->etime = carve(&cur, (size_t)n * 8192);
  g->etime_r = carve(&cur, (size_t)n * 8192);
  g->edf = carve(&cur, sizeof(xaac_hbe_dft_state) * n);
  g->ecf = carve(&cur, sizeof(xaac_hbe_dft_cfg) * n);
  g->ecoef = carve(&cur, (size_t)n * 2 * 64 * 128 * 4);
  g->ecfg = carve(&cur, (size_t)n * 4);
}

/* ---- child side: post the request, sleep until the parent has the results in the staging row ------------------- */
static void rendezvous(int req) {
  inst_t *me = &my_grp->inst[my_idx];
  me->req = req;
  me->calls[req]++;
  atomic_fetch_add(&my_grp->pending, 1);
  sem_post(&g_hdr->wake);
  while (sem_wait(&me->done) != 0 && errno == EINTR) {
  }
}

VOID __real_ixheaacd_imdct_process(ia_aac_dec_overlap_info *, WORD32 *, ia_ics_info_struct *, VOID *, const WORD16,
                                   WORD32 *, ia_aac_dec_tables_struct *, WORD32, WORD32, WORD);
VOID __wrap_ixheaacd_imdct_process(ia_aac_dec_overlap_info *oi, WORD32 *spec, ia_ics_info_struct *ics, VOID *out,
                                   const WORD16 ch_fac, WORD32 *scratch, ia_aac_dec_tables_struct *tabs,
                                   WORD32 object_type, WORD32 ld_mps_present, WORD slot_element) {
  group_t *g = my_grp;
  const int i = my_idx;
  if (!g || ics->frame_length != 1024 || ld_mps_present || object_type == AOT_ER_AAC_LD || object_type == AOT_ER_AAC_ELD) {
    __real_ixheaacd_imdct_process(oi, spec, ics, out, ch_fac, scratch, tabs, object_type, ld_mps_present, slot_element);
    return;
  }
  memcpy(g->spec + 1024 * (size_t)i, spec, 4096);
  memcpy(g->ovl + 512 * (size_t)i, oi->ptr_overlap_buf, 2048);
  g->ics[i].window_sequence = (uint8_t)ics->window_sequence;
  g->ics[i].window_shape = (uint8_t)ics->window_shape;
  g->ost[i].window_sequence = (uint8_t)oi->window_sequence;
  g->ost[i].window_shape = (uint8_t)oi->window_shape;
  rendezvous(REQ_IMDCT);
  {
    const int32_t *o = g->out32 + 1024 * (size_t)i;
    for (int k = 0; k < 1024; k++) ((WORD32 *)out)[k * ch_fac] = o[k];
  }
  memcpy(oi->ptr_overlap_buf, g->ovl + 512 * (size_t)i, 2048);
  oi->window_sequence = g->ost[i].window_sequence;
  oi->window_shape = g->ost[i].window_shape;
  ics->qshift_adj = g->qadj[i];
}

/* ---- -esbr_hq:1: the DFT transposer's struct members <-> the group's rows (as oracle/ref_dropin.c: dft_upload / dft_download) ---- */
static int dft_sizes_fit(const ia_esbr_hbe_txposer_struct *t) {
  const int ana0 = t->ana_fft_size[0], syn0 = t->syn_fft_size[0];
  return ana0 >= 0 && ana0 <= XAAC_HBE_DFT_MAX_ANA && syn0 >= 0 && syn0 <= XAAC_HBE_DFT_MAX_SYN && ana0 == 32 * t->synth_size &&
         syn0 == 16 * t->analy_size;
}
static void dft_to_rows(const ia_esbr_hbe_txposer_struct *t, group_t *g, int i) {
  xaac_hbe_dft_state *st = &g->edf[i];
  xaac_hbe_dft_cfg *cfg = &g->ecf[i];
  const int ana0 = t->ana_fft_size[0], syn0 = t->syn_fft_size[0];
  int tr, o;
  memset(st, 0, sizeof(*st));
  memcpy(st->input_buf, t->ptr_input_buf, sizeof(float) * 2 * ana0);
  memcpy(st->output_buf, t->ptr_output_buf, sizeof(float) * 4 * syn0);
  memcpy(st->synth_buf, t->synth_buf, sizeof(st->synth_buf));
  memcpy(st->anal.analy_buf, t->analy_buf, sizeof(st->anal.analy_buf));
  st->anal.analy_size = t->analy_size;
  st->anal.a_start = t->a_start;
  st->synth_size = t->synth_size;
  st->k_start = t->k_start;
  st->start_band = t->start_band;
  st->end_band = t->end_band;
  st->max_stretch = t->max_stretch;
  for (o = 0; o < 6; o++) st->x_over_qmf[o] = t->x_over_qmf[o];
  memset(cfg, 0, sizeof(*cfg));
  memcpy(cfg->anal_window, t->anal_window, sizeof(float) * ana0);
  memcpy(cfg->synth_window, t->synth_window, sizeof(float) * syn0);
  for (tr = 0; tr < 3; tr++)
    for (o = 0; o < 2; o++) memcpy(cfg->fd_win[tr][o], t->fd_win_buf[tr][o], sizeof(cfg->fd_win[tr][o]));
  memcpy(g->ecoef + (size_t)i * 64 * 128, t->str_dft_hbe_anal_coeff.real, 64 * 128 * 4);
  memcpy(g->ecoef + ((size_t)g->n + i) * 64 * 128, t->str_dft_hbe_anal_coeff.imag, 64 * 128 * 4);
  g->ecfg[i] = i;
}
static void dft_from_rows(const group_t *g, int i, ia_esbr_hbe_txposer_struct *t) {
  const xaac_hbe_dft_state *st = &g->edf[i];
  const int ana0 = t->ana_fft_size[0], syn0 = t->syn_fft_size[0];
  if (st->last_status) return;
  memcpy(t->ptr_input_buf, st->input_buf, sizeof(float) * 2 * ana0);
  memcpy(t->ptr_output_buf, st->output_buf, sizeof(float) * 4 * syn0);
  memcpy(t->synth_buf, st->synth_buf, sizeof(st->synth_buf));
  memcpy(t->analy_buf, st->anal.analy_buf, sizeof(st->anal.analy_buf));
}

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
  group_t *g = my_grp;
  const int i = my_idx;
  const int with_ps = !low_pow && ps && h->channel_mode == PS_STEREO;
  const int ps_on = with_ps && apply;
  /* the reference's default flags: the eSBR branch (sbr_dec.c:816-1009) of an HE-AAC channel / HE-AACv2 stream, as in
     oracle/ref_dropin.c */
  if (g && h->enh_sbr && aot != AOT_ER_AAC_ELD && aot != AOT_ER_AAC_LD && !h->usac_flag && h->hbe_flag && d->p_hbe_txposer != NULL &&
      (!h->esbr_hq || dft_sizes_fit(d->p_hbe_txposer)) &&
      (h->channel_mode == PS_STEREO ? (ps != NULL && synth_r != NULL && !ps->use_34_st_bands && !ps->use_pca_rot_flg && ps->ps_mode == 0)
                                    : !h->enh_sbr_ps) &&
      !drc_on && !ldmps && !mps && !f->mps_sbr_flag && f->sbr_mode != PVC_SBR && h->sbr_ratio_idx != SBR_UPSAMPLE_IDX_4_1 &&
      h->num_time_slots == 16 && d->str_codec_qmf_bank.no_channels == 32 &&
      d->str_synthesis_qmf_bank.no_channels == 64) {
    const ia_qmf_dec_tables_struct *q = tabs->qmf_dec_tables_ptr;
    const int eps = h->channel_mode == PS_STEREO;
    d->str_synthesis_qmf_bank.filter_pos_syn_32 += q->esbr_qmf_c - d->str_synthesis_qmf_bank.p_filter_32;
    d->str_synthesis_qmf_bank.p_filter_32 = q->esbr_qmf_c;
    if (eps) {
      synth_r->filter_pos_syn_32 += q->esbr_qmf_c - synth_r->p_filter_32;
      synth_r->p_filter_32 = q->esbr_qmf_c;
    }
    to_header(h, d, &g->hdr[i]);
    to_frame(f, apply, &g->frm[i]);
    to_esbr_side(h, f, &g->esd[i]);
    to_esbr_state(d, h, f, &g->est[i]);
    g->inst[i].dft = h->esbr_hq ? 1 : 0;
    if (h->esbr_hq) dft_to_rows(d->p_hbe_txposer, g, i);
    else to_hbe_state(d->p_hbe_txposer, &g->ehb[i]);
    memcpy(g->ecore + 1024 * (size_t)i, d->time_sample_buf, 4096);
    if (eps) {
      to_ps_frame(ps, &g->psf[i]);
      to_esbr_ps_state(ps, synth_r, &g->eps[i]);
    }
    rendezvous(eps ? REQ_ESBR_PS : REQ_ESBR);
    if (g->status[i]) return g->status[i];
    memcpy(d->time_sample_buf, g->etime + 2048 * (size_t)i, 8192);
    from_esbr_state(&g->est[i], d, h, f);
    if (apply && h->esbr_hq) dft_from_rows(g, i, d->p_hbe_txposer);
    else if (apply) from_hbe_state(&g->ehb[i], d->p_hbe_txposer, h);
    if (eps) {
      memcpy(ps->time_sample_buf[1], g->etime_r + 2048 * (size_t)i, 8192);
      from_esbr_ps_state(&g->eps[i], ps, synth_r);
      ps->use_34_st_bands_prev = ps->use_34_st_bands;
      ((ia_sbr_frame_info_data_struct *)((ia_handle_sbr_dec_inst_struct)self)->frame_buffer[1])->reset_flag = 0;
    }
    d->band_count = h->pstr_freq_band_data->sub_band_end;
    f->reset_flag = 0;
    f->prev_sbr_mode = f->sbr_mode;
    return 0;
  }
  if (!g || h->enh_sbr || aot == AOT_ER_AAC_ELD || aot == AOT_ER_AAC_LD || drc_on || ldmps || mps ||
      h->num_time_slots * h->time_step != 32)
    return __real_ixheaacd_sbr_dec(d, time_data, h, f, p, ps, synth_r, sf_r, apply, low_pow, work, tabs, common, ch_fac,
                                   pvc, drc_on, drc, aot, ldmps, self, mps, ec);
  to_header(h, d, &g->hdr[i]);
  to_frame(f, apply, &g->frm[i]);
  to_state(d, p, low_pow, &g->sst[i]);
  {
    int16_t *in = g->pin + 1024 * (size_t)i;
    for (int k = 0; k < 1024; k++) in[k] = time_data[k * ch_fac];
  }
  if (with_ps) {
    to_ps_frame(ps, &g->psf[i]);
    to_ps_state(ps, synth_r, sf_r, &g->pss[i]);
  }
  rendezvous(low_pow ? REQ_SBR_LP : (with_ps ? REQ_SBR_PS : REQ_SBR_HQ));
  if (g->status[i]) return g->status[i]; /* the reference returns before touching anything */
  from_state(&g->sst[i], low_pow, d, p);
  {
    const int16_t *o = g->pout + (with_ps ? 4096 : 2048) * (size_t)i; /* L,R pairs with PS, else 2048 samples: dense rows */
    if (with_ps) {
      from_ps_state(&g->pss[i], ps, synth_r, sf_r);
      for (int k = 0; k < 2048; k++) {
        time_data[k * ch_fac] = o[2 * k];
        if (ps_on) time_data[k * ch_fac + 1] = o[2 * k + 1];
      }
    } else {
      for (int k = 0; k < 2048; k++) time_data[k * ch_fac] = o[k];
    }
  }
  return 0;
}

VOID __real_ixheaacd_peak_limiter_process(ia_peak_limiter_struct *, VOID *, UWORD32, UWORD8 *);
VOID __wrap_ixheaacd_peak_limiter_process(ia_peak_limiter_struct *lim, VOID *samples, UWORD32 frame_len,
                                          UWORD8 *qshift_adj) {
  group_t *g = my_grp;
  const int i = my_idx;
  const UWORD32 nch = lim->num_channels;
  if (!g || nch < 1 || nch > LIM_MAX_CH || lim->attack_time_samples < 1 || lim->attack_time_samples > XAAC_LIM_MAX_ATTACK ||
      frame_len != 1024) {
    __real_ixheaacd_peak_limiter_process(lim, samples, frame_len, qshift_adj);
    return;
  }
  xaac_limiter_state *st = &g->lst[i];
  memset(st, 0, sizeof(*st));
  ref_limiter_from_ref(st, lim);
  memcpy(g->lx + (size_t)i * 1024 * LIM_MAX_CH, samples, (size_t)frame_len * nch * 4);
  memcpy(g->lq + (size_t)i * nch, qshift_adj, nch); /* rows num_channels apart, as the ABI wants them */
  g->inst[i].lim_nch = (int)nch;
  g->inst[i].lim_len = (int)frame_len;
  rendezvous(REQ_LIM);
  memcpy(samples, g->lx + (size_t)i * 1024 * LIM_MAX_CH, (size_t)frame_len * nch * 4);
  { /* the reference keeps its buffer pointers; only the contents come back */
    const UWORD32 a = lim->attack_time_samples;
    lim->gain_modified = st->gain_modified;
    lim->pre_smoothed_gain = st->pre_smoothed_gain;
    lim->delayed_input_index = st->delayed_input_index;
    lim->min_gain = st->min_gain;
    lim->max_idx = st->max_idx;
    lim->cir_buf_pnt = st->cir_buf_pnt;
    memcpy(lim->max_buf, st->max_buf, a * sizeof(FLOAT32));
    memcpy(lim->delayed_input, st->delayed_input, (size_t)a * nch * sizeof(FLOAT32));
  }
}

/* ---- parent side --------------------------------------------------------------------------------------------- */
typedef struct {
  hipStream_t stream;
  hipEvent_t ev;
  xaac_ctx *ctx;
  int inflight, req;
  char *dev; /* device mirror of the staging arrays, same layout */
  group_t d; /* device pointers */
  void *ws;
  uint64_t ws_bytes;
} gpu_group_t;

static void die(const char *what) {
  fprintf(stderr, "xaacdec_batch: %s failed\n", what);
  kill(0, SIGTERM);
  exit(3);
}
#define HIP(x) do { if ((x) != hipSuccess) die(#x); } while (0)
#define H2D(field, bytes) HIP(hipMemcpyAsync(gg->d.field, g->field, (bytes), hipMemcpyHostToDevice, gg->stream))
#define D2H(field, bytes) HIP(hipMemcpyAsync(g->field, gg->d.field, (bytes), hipMemcpyDeviceToHost, gg->stream))

/* all live instances of the group wait at a seam: one batch for each kind of request present (one kind, unless the
   streams of a group differ) */
static long g_batches_dft; /* eSBR batches that carried DFT transposers */
static void launch_group(group_t *g, gpu_group_t *gg, long *batches) {
  const size_t n = (size_t)g->n;
  int kinds[REQ_KINDS] = {0};
  for (int i = 0; i < g->n; i++) kinds[g->inst[i].req]++;
  for (int req = REQ_IMDCT; req < REQ_KINDS; req++) {
    if (!kinds[req]) continue;
    /* rows that are not part of this batch (instances that have finished, or wait with another kind of request) are
       processed along harmlessly: their rows hold stale but well-formed operands and nobody reads the results */
    batches[req]++;
    if (req == REQ_IMDCT) {
      xaac_imdct_batch b;
      memset(&b, 0, sizeof(b));
      H2D(spec, n * 4096); H2D(ovl, n * 2048); H2D(ics, n * sizeof(xaac_ics_info)); H2D(ost, n * sizeof(xaac_ovl_state));
      b.n_ch = g->n; b.ch_fac = 1; b.spec = gg->d.spec; b.ics = gg->d.ics; b.overlap = gg->d.ovl; b.state = gg->d.ost;
      b.out32 = gg->d.out32; b.qshift_adj = gg->d.qadj; b.pcm_mode = XAAC_PCM_LC;
      if (xaac_imdct_process_batch(gg->ctx, &b) != XAAC_OK) die("xaac_imdct_process_batch");
      D2H(out32, n * 4096); D2H(ovl, n * 2048); D2H(ost, n * sizeof(xaac_ovl_state)); D2H(qadj, n);
    } else if (req == REQ_LIM) {
      xaac_limiter_batch b;
      int nch = 0, len = 0;
      for (int i = 0; i < g->n; i++)
        if (g->inst[i].req == REQ_LIM) { nch = g->inst[i].lim_nch; len = g->inst[i].lim_len; }
      memset(&b, 0, sizeof(b));
      H2D(lx, n * 1024 * LIM_MAX_CH * 4); H2D(lq, n * LIM_MAX_CH); H2D(lst, n * sizeof(xaac_limiter_state));
      b.n_streams = g->n; b.frame_len = len; b.num_channels = nch; b.samples = gg->d.lx; b.stride = 1024 * LIM_MAX_CH;
      b.qshift_adj = gg->d.lq; b.state = gg->d.lst; b.workspace = gg->ws; b.workspace_bytes = gg->ws_bytes;
      if (xaac_peak_limiter_process_batch(gg->ctx, &b) != XAAC_OK) die("xaac_peak_limiter_process_batch");
      D2H(lx, n * 1024 * LIM_MAX_CH * 4); D2H(lst, n * sizeof(xaac_limiter_state));
    } else if (req == REQ_ESBR || req == REQ_ESBR_PS) {
      xaac_esbr_sbr_batch b;
      memset(&b, 0, sizeof(b));
      H2D(hdr, n * sizeof(xaac_sbr_header)); H2D(frm, n * sizeof(xaac_sbr_frame)); H2D(esd, n * sizeof(xaac_esbr_side));
      int dft = 0; /* -esbr_hq:1 is the run's flag: the instances of a rendezvous agree on it */
      for (int i = 0; i < g->n; i++)
        if (g->inst[i].req == req && g->inst[i].dft) dft = 1;
      H2D(est, n * sizeof(xaac_esbr_state)); H2D(ecore, n * 4096);
      if (dft) {
        H2D(edf, n * sizeof(xaac_hbe_dft_state)); H2D(ecf, n * sizeof(xaac_hbe_dft_cfg)); H2D(ecoef, n * 2 * 64 * 128 * 4); H2D(ecfg, n * 4);
      } else {
        H2D(ehb, n * sizeof(xaac_hbe_state));
      }
      b.n_ch = g->n; b.core = gg->d.ecore; b.header = gg->d.hdr; b.frame = gg->d.frm; b.side = gg->d.esd; b.state = gg->d.est;
      b.out = gg->d.etime; b.status = gg->d.status; b.workspace = gg->ws; b.workspace_bytes = gg->ws_bytes;
      if (dft) { /* instance i's configuration is number i */
        b.hbe_dft_state = gg->d.edf; b.hbe_dft_cfg_tab = gg->d.ecf; b.hbe_dft_cfg = gg->d.ecfg;
        b.hbe_dft_coef_re = gg->d.ecoef; b.hbe_dft_coef_im = gg->d.ecoef + n * 64 * 128;
        g_batches_dft++;
      } else {
        b.hbe_state = gg->d.ehb;
      }
      if (req == REQ_ESBR_PS) {
        H2D(psf, n * sizeof(xaac_ps_frame)); H2D(eps, n * sizeof(xaac_esbr_ps_state));
        b.ps_frame = gg->d.psf; b.ps_state = gg->d.eps; b.out_r = gg->d.etime_r;
      }
      if (xaac_esbr_sbr_process_batch(gg->ctx, &b) != XAAC_OK) die("xaac_esbr_sbr_process_batch");
      D2H(etime, n * 8192); D2H(est, n * sizeof(xaac_esbr_state)); D2H(status, n * 4);
      if (dft) D2H(edf, n * sizeof(xaac_hbe_dft_state));
      else D2H(ehb, n * sizeof(xaac_hbe_state));
      if (req == REQ_ESBR_PS) { D2H(etime_r, n * 8192); D2H(eps, n * sizeof(xaac_esbr_ps_state)); }
    } else {
      H2D(hdr, n * sizeof(xaac_sbr_header)); H2D(frm, n * sizeof(xaac_sbr_frame)); H2D(sst, n * sizeof(xaac_sbr_state));
      H2D(pin, n * 2048);
      if (req == REQ_SBR_LP) {
        xaac_sbr_lp_batch b;
        memset(&b, 0, sizeof(b));
        b.n_ch = g->n; b.in_ch_fac = b.out_ch_fac = 1; b.pcm_in = gg->d.pin; b.header = gg->d.hdr; b.frame = gg->d.frm;
        b.state = gg->d.sst; b.pcm_out = gg->d.pout; b.status = gg->d.status; b.workspace = gg->ws;
        b.workspace_bytes = gg->ws_bytes;
        if (xaac_sbr_lp_process_batch(gg->ctx, &b) != XAAC_OK) die("xaac_sbr_lp_process_batch");
        D2H(pout, n * 4096); /* 2048 samples per channel, dense */
      } else {
        xaac_sbr_hq_batch b;
        memset(&b, 0, sizeof(b));
        b.n_ch = g->n; b.in_ch_fac = b.out_ch_fac = 1; b.pcm_in = gg->d.pin; b.header = gg->d.hdr; b.frame = gg->d.frm;
        b.state = gg->d.sst; b.pcm_out = gg->d.pout; b.status = gg->d.status; b.workspace = gg->ws;
        b.workspace_bytes = gg->ws_bytes;
        if (req == REQ_SBR_PS) {
          H2D(psf, n * sizeof(xaac_ps_frame)); H2D(pss, n * sizeof(xaac_ps_state));
          b.ps_frame = gg->d.psf; b.ps_state = gg->d.pss;
        }
        if (xaac_sbr_hq_process_batch(gg->ctx, &b) != XAAC_OK) die("xaac_sbr_hq_process_batch");
        if (req == REQ_SBR_PS) {
          D2H(pss, n * sizeof(xaac_ps_state));
          D2H(pout, n * 8192);
        } else {
          D2H(pout, n * 4096); /* mono HQ: 2048 samples per stream, dense */
        }
      }
      D2H(sst, n * sizeof(xaac_sbr_state)); D2H(status, n * 4);
    }
  }
  HIP(hipEventRecord(gg->ev, gg->stream));
  gg->inflight = 1;
}

static void release_group(group_t *g, gpu_group_t *gg) {
  gg->inflight = 0;
  atomic_store(&g->pending, 0);
  for (int i = 0; i < g->n; i++)
    if (g->inst[i].req != REQ_NONE) {
      g->inst[i].req = REQ_NONE;
      sem_post(&g->inst[i].done);
    }
}

int main(int argc, char **argv) {
  int first_group = -1, n_groups, total = 0;
  for (int i = 1; i < argc; i++)
    if (!strcmp(argv[i], "--")) first_group = i + 1;
  if (first_group < 0 || first_group >= argc) {
    fprintf(stderr, "usage: %s [decoder flags] -- N:input.aac:output-prefix ...\n", argv[0]);
    return 2;
  }
  n_groups = argc - first_group;
  int *gn = calloc(n_groups, sizeof(int));
  char **gin = calloc(n_groups, sizeof(char *)), **gout = calloc(n_groups, sizeof(char *));
  size_t bytes = 4096 + sizeof(group_t) * n_groups;
  for (int k = 0; k < n_groups; k++) {
    char *spec = strdup(argv[first_group + k]);
    char *c1 = strchr(spec, ':'), *c2 = c1 ? strchr(c1 + 1, ':') : NULL;
    if (!c2) { fprintf(stderr, "bad group '%s'\n", spec); return 2; }
    *c1 = *c2 = 0;
    gn[k] = atoi(spec); gin[k] = c1 + 1; gout[k] = c2 + 1;
    if (gn[k] < 1) return 2;
    bytes += group_bytes(gn[k]);
    total += gn[k];
  }
  bytes = (bytes + 4095) & ~(size_t)4095;
  char *base = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (base == MAP_FAILED) { perror("mmap"); return 3; }
  char *cur = base;
  g_hdr = carve(&cur, sizeof(shared_hdr_t));
  g_grp = carve(&cur, sizeof(group_t) * n_groups);
  char **gbase = calloc(n_groups, sizeof(char *));
  sem_init(&g_hdr->wake, 1, 0);
  g_hdr->n_groups = n_groups;
  for (int k = 0; k < n_groups; k++) {
    gbase[k] = cur;
    group_layout(&g_grp[k], gn[k], cur);
    cur += group_bytes(gn[k]);
    atomic_init(&g_grp[k].pending, 0);
    atomic_init(&g_grp[k].live, gn[k]);
    for (int i = 0; i < gn[k]; i++) sem_init(&g_grp[k].inst[i].done, 1, 0);
  }
  /* ---- the decoder instances: forked BEFORE anything touches HIP ---- */
  pid_t *pids = calloc(total, sizeof(pid_t));
  int np = 0;
  for (int k = 0; k < n_groups; k++)
    for (int i = 0; i < gn[k]; i++) {
      pid_t pid = fork();
      if (pid < 0) { perror("fork"); kill(0, SIGTERM); return 3; }
      if (pid == 0) {
        char a_in[4096], a_out[4096];
        char *cargv[64];
        int cargc = 0;
        my_grp = &g_grp[k];
        my_idx = i;
        snprintf(a_in, sizeof(a_in), "-ifile:%s", gin[k]);
        snprintf(a_out, sizeof(a_out), "-ofile:%s.%d.wav", gout[k], i);
        cargv[cargc++] = argv[0];
        cargv[cargc++] = a_in;
        cargv[cargc++] = a_out;
        for (int a = 1; a < first_group - 1 && cargc < 63; a++) cargv[cargc++] = argv[a];
        cargv[cargc] = NULL;
        { /* the test bench is chatty */
          int nul = open("/dev/null", O_WRONLY);
          if (nul >= 0) { dup2(nul, 1); close(nul); }
        }
        int rc = ref_cli_main(cargc, cargv);
        fflush(NULL);
        atomic_fetch_sub(&my_grp->live, 1);
        sem_post(&g_hdr->wake);
        _exit(rc ? 4 : 0);
      }
      pids[np++] = pid;
    }
  /* ---- parent: the only process with a HIP context ---- */
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  gpu_group_t *gg = calloc(n_groups, sizeof(gpu_group_t));
  int pinned = hipHostRegister(base, bytes, hipHostRegisterDefault) == hipSuccess;
  for (int k = 0; k < n_groups; k++) {
    HIP(hipStreamCreateWithFlags(&gg[k].stream, hipStreamNonBlocking));
    HIP(hipEventCreateWithFlags(&gg[k].ev, hipEventDisableTiming));
    if (xaac_create(&gg[k].ctx, 0, gg[k].stream) != XAAC_OK) die("xaac_create");
    HIP(hipMalloc((void **)&gg[k].dev, group_bytes(gn[k])));
    group_layout(&gg[k].d, gn[k], gg[k].dev);
    uint64_t a = xaac_sbr_lp_workspace_bytes(gn[k]), b = xaac_sbr_hq_workspace_bytes(gn[k], 1),
             c = xaac_peak_limiter_workspace_bytes(gn[k]);
    gg[k].ws_bytes = a > b ? (a > c ? a : c) : (b > c ? b : c);
    if (xaac_esbr_workspace_bytes(gn[k]) > gg[k].ws_bytes) gg[k].ws_bytes = xaac_esbr_workspace_bytes(gn[k]);
    HIP(hipMalloc(&gg[k].ws, gg[k].ws_bytes));
  }
  long batches[REQ_KINDS] = {0};
  for (;;) {
    int alive = 0, progressed = 0;
    for (int k = 0; k < n_groups; k++) {
      group_t *g = &g_grp[k];
      const int live = atomic_load(&g->live);
      alive += live;
      if (gg[k].inflight) {
        hipError_t e = hipEventQuery(gg[k].ev);
        if (e == hipSuccess) { release_group(g, &gg[k]); progressed = 1; }
        else if (e != hipErrorNotReady) die("hipEventQuery");
      } else if (live > 0 && atomic_load(&g->pending) == live) {
        launch_group(g, &gg[k], batches);
        progressed = 1;
      }
    }
    if (!alive) break;
    if (!progressed) {
      int busy = 0;
      for (int k = 0; k < n_groups; k++) busy |= gg[k].inflight;
      if (busy) sched_yield(); /* a batch is on the GPU: poll its event */
      else {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        ts.tv_nsec += 2000000;
        if (ts.tv_nsec >= 1000000000) { ts.tv_sec++; ts.tv_nsec -= 1000000000; }
        sem_timedwait(&g_hdr->wake, &ts);
        for (int i = 0; i < np; i++) { /* an instance that died at a seam would leave its group waiting for ever */
          int st = 0;
          if (pids[i] > 0 && waitpid(pids[i], &st, WNOHANG) == pids[i]) {
            if (!(WIFEXITED(st) && WEXITSTATUS(st) == 0)) die("a decoder instance");
            pids[i] = -pids[i];
          }
        }
      }
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  int failed = 0;
  for (int i = 0; i < np; i++) {
    int st = 0;
    if (pids[i] < 0) continue; /* reaped above, exit status 0 */
    waitpid(pids[i], &st, 0);
    failed += !(WIFEXITED(st) && WEXITSTATUS(st) == 0);
  }
  long calls[REQ_KINDS] = {0};
  for (int k = 0; k < n_groups; k++)
    for (int i = 0; i < gn[k]; i++)
      for (int r = 0; r < REQ_KINDS; r++) calls[r] += g_grp[k].inst[i].calls[r];
  printf("{\"streams\": %d, \"groups\": %d, \"failed\": %d, \"pinned\": %d, \"seconds\": %.6f, "
         "\"calls\": {\"imdct\": %ld, \"sbr_lp\": %ld, \"sbr_hq\": %ld, \"sbr_ps\": %ld, \"limiter\": %ld, \"esbr\": %ld, \"esbr_ps\": %ld}, "
         "\"batches\": {\"imdct\": %ld, \"sbr_lp\": %ld, \"sbr_hq\": %ld, \"sbr_ps\": %ld, \"limiter\": %ld, \"esbr\": %ld, \"esbr_ps\": %ld, \"esbr_with_dft_transposer\": %ld}}\n",
         total, n_groups, failed, pinned, (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec), calls[REQ_IMDCT],
         calls[REQ_SBR_LP], calls[REQ_SBR_HQ], calls[REQ_SBR_PS], calls[REQ_LIM], calls[REQ_ESBR], calls[REQ_ESBR_PS],
         batches[REQ_IMDCT], batches[REQ_SBR_LP], batches[REQ_SBR_HQ], batches[REQ_SBR_PS], batches[REQ_LIM], batches[REQ_ESBR],
         batches[REQ_ESBR_PS], g_batches_dft);
  return failed ? 1 : 0;
}
