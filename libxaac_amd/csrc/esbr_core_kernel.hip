/*
 * esbr_core_kernel.hip -- gfx950 kernel for the float middle of the reference's default SBR path (Path A) on HE-AAC
 * channels: ixheaacd_generate_hf (sbrdec_lpfuncs.c:981), ixheaacd_sbr_env_calc (esbr_envcal.c:71), the regrouping of
 * ixheaacd_esbr_synthesis_regrp (sbr_dec.c:297) and the history shifts of sbr_dec.c:835-857; arithmetic in esbr_core.h.
 *
 * Mapping: one wave = one channel-frame, lane = QMF band.  The path works 38 slots behind the analysis bank (op_delay 6 +
 * the 32 slots the reference reserves for its harmonic transposer), so the HF generator reads the channel's 40-row
 * history straight from its state; this frame's analysis rows only enter the history at the end.  sbr_qmf_out lives in a
 * 42-row global scratch (L2-resident while the wave works on it), side info and per-band vectors in LDS.  The stages are
 * chains of short dependent loops, so the kernel needs many resident waves AND few exposed memory latencies: column walks
 * move eight rows per burst (esbr_core.h: XE_CH), the copies eight rows per burst, the side info is staged in LDS for the
 * scalar table walks.  With the matrix in LDS (21.5 KB, 6 waves per CU) the same code measured 1.4x slower, twice.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef XE_PROFILE /* tools/prof_esbr_core.py: cycles of lane 0 between XE_T hooks, summed over channel-frames */
__shared__ long long xe_prof_last;
__shared__ long long xe_prof_acc[16];
#define XE_T(i)                            \
  do {                                     \
    if (threadIdx.x == 0) {                \
      const long long t_ = clock64();      \
      xe_prof_acc[i] += t_ - xe_prof_last; \
      xe_prof_last = t_;                   \
    }                                      \
  } while (0)
#endif
/* LDS copy of the 512 random phases the noise substitution looks up per band and slot */
__shared__ float xe_lds_random_phase[1024];
#define XE_RANDOM_PHASE(i) xe_lds_random_phase[i]
/* rows a column walk keeps in flight (esbr_core.h: XE_CH): twelve measured best for this kernel (8: 413 us, 12: 398, 16: 440 with
   70 spills); the float PS kernel stays at eight (twelve: 320 -> 350 us) */
#define XE_CH 12
#include "esbr_core.h"
#include "pvc.h"       /* xp_process: the PVC decoder of a USAC channel's PVC frames, run in this kernel */
#include "hbe_trans.h" /* xh_apply_params_ok */
#include "hbe_kernel.h" /* XAAC_HBE_LDS_OK */
#include "esbr_core_kernel.h"

namespace {
/* global -> LDS: eight loads are in flight before the first store */
__device__ __forceinline__ void xe_copy_words(int32_t *dst, const int32_t *src, int n, int lane) {
  for (int i = lane; i < n; i += 64 * 8) {
    int32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (i + 64 * j < n) t[j] = src[i + 64 * j];
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (i + 64 * j < n) dst[i + 64 * j] = t[j];
  }
}
}  // namespace

/* HARM: with the harmonic transposer's rows (batches that hand in hbe_state); the other variant carries none of that code */
/* PVC: channels whose host tracks the PVC side info and state (USAC; xaac_esbr.h): PVC frames decoded and adjusted here */
/* USF4: 4:1 SBR (USAC channels without a transposer): 64 slots, four to an envelope time slot; the low-band matrix is the
   80-row scratch the 16-channel analysis bank wrote rows 8..71 of, not the channel's 40-row state */
template <bool HARM, bool PVC = false, bool USF4 = false>
__global__ __launch_bounds__(64, USF4 ? 2 : 3) void xaac_esbr_core_kernel(XaacEsbrCoreParams p) { /* (4:1: the 14-row histories in flight take the registers of a two-wave budget) */ /* 168 VGPRs: 12 waves per CU (measured best of 8 / 12 / 16) */
#ifdef XE_PROFILE
  if (threadIdx.x == 0) {
    for (int i = 0; i < 16; i++) xe_prof_acc[i] = 0;
    xe_prof_last = clock64();
  }
#endif
  __shared__ XeWork w;
  const int ch = blockIdx.x, lane = threadIdx.x;
  const XsCx cx = {lane, 64};
  /* the side info is walked by scalar, data-dependent code (band tables, patch construction): from LDS, not from global */
  __shared__ xaac_sbr_header sh;
  __shared__ xaac_sbr_frame sf;
  __shared__ xaac_esbr_side ssd;
  static_assert(sizeof(xaac_sbr_header) % 4 == 0 && sizeof(xaac_sbr_frame) % 4 == 0 && sizeof(xaac_esbr_side) % 4 == 0, "word copies");
  xaac_esbr_state *st = p.state + ch;
  constexpr int RATE = USF4 ? 4 : 2, SLOTS = USF4 ? 64 : 32;
  constexpr int OUT_ROWS = USF4 ? XAAC_ESBR_OUT_ROWS_4_1 : XAAC_ESBR_OUT_ROWS, L_ROWS = USF4 ? 64 : XAAC_ESBR_L_ROWS;
  const float *are = p.ana_re + (size_t)ch * 2048, *aim = p.ana_im + (size_t)ch * 2048;
  /* the low-band matrix the tools read (qmf_buf_real / _imag from the reference's row 0) */
  float *qre = USF4 ? p.q_re + (size_t)ch * XAAC_ESBR_Q_ROWS_4_1 * 64 : &st->qmf_re[0][0];
  float *qim = USF4 ? p.q_im + (size_t)ch * XAAC_ESBR_Q_ROWS_4_1 * 64 : &st->qmf_im[0][0];
  constexpr int HIST = USF4 ? XAAC_ESBR_OUT_HIST_ROWS_4_1 : XAAC_ESBR_OUT_HIST_ROWS; /* op_delay + 2 rows of both matrices carried */
  float hist_re[HIST], hist_im[HIST]; /* sbr_qmf_out's rows of history: fetched with the side info, stored below */
  { /* header, frame, side info and the random-phase table: every load in flight before the first LDS store -- one memory
       latency for the lot (as four copies one behind the other they were four, and the history rows a fifth) */
    constexpr int NH = (sizeof(sh) / 4 + 63) / 64, NF = (sizeof(sf) / 4 + 63) / 64, NS = (sizeof(ssd) / 4 + 63) / 64, NR = 1024 / 64;
    const int32_t *gh = reinterpret_cast<const int32_t *>(p.header + ch), *gf = reinterpret_cast<const int32_t *>(p.frame + ch);
    const int32_t *gs = reinterpret_cast<const int32_t *>(p.side + ch), *gr = reinterpret_cast<const int32_t *>(xaac_esbr_random_phase);
    int32_t th[NH], tf[NF], ts[NS], tr[NR];
#pragma unroll
    for (int j = 0; j < NH; j++) th[j] = lane + 64 * j < (int)(sizeof(sh) / 4) ? gh[lane + 64 * j] : 0;
#pragma unroll
    for (int j = 0; j < NF; j++) tf[j] = lane + 64 * j < (int)(sizeof(sf) / 4) ? gf[lane + 64 * j] : 0;
#pragma unroll
    for (int j = 0; j < NS; j++) ts[j] = lane + 64 * j < (int)(sizeof(ssd) / 4) ? gs[lane + 64 * j] : 0;
#pragma unroll
    for (int j = 0; j < NR; j++) tr[j] = gr[lane + 64 * j];
#pragma unroll
    for (int r = 0; r < HIST; r++) { /* (4:1: rows 8..13 ride in the state's ph rows, xaac_esbr.h) */
      hist_re[r] = r < 8 ? st->out_re[r][lane] : st->ph_re[r - 8][lane];
      hist_im[r] = r < 8 ? st->out_im[r][lane] : st->ph_im[r - 8][lane];
    }
#pragma unroll
    for (int j = 0; j < NH; j++)
      if (lane + 64 * j < (int)(sizeof(sh) / 4)) reinterpret_cast<int32_t *>(&sh)[lane + 64 * j] = th[j];
#pragma unroll
    for (int j = 0; j < NF; j++)
      if (lane + 64 * j < (int)(sizeof(sf) / 4)) reinterpret_cast<int32_t *>(&sf)[lane + 64 * j] = tf[j];
#pragma unroll
    for (int j = 0; j < NS; j++)
      if (lane + 64 * j < (int)(sizeof(ssd) / 4)) reinterpret_cast<int32_t *>(&ssd)[lane + 64 * j] = ts[j];
#pragma unroll
    for (int j = 0; j < NR; j++) reinterpret_cast<int32_t *>(xe_lds_random_phase)[lane + 64 * j] = tr[j];
  }
  __syncthreads();
  const xaac_sbr_header *h = &sh;
  const xaac_sbr_frame *f = &sf;
  const xaac_esbr_side *sd = &ssd;
  XE_T(0);
  float *ore = p.out_re + (size_t)ch * OUT_ROWS * 64, *oim = p.out_im + (size_t)ch * OUT_ROWS * 64;
  float *rre = p.syn_re + (size_t)ch * L_ROWS * 64, *rim = p.syn_im + (size_t)ch * L_ROWS * 64;
  const int apply = f->apply_processing != 0;
  int rc = 0;
  if (apply && xe_side_info_bad(h, f, sd)) rc = -1;
  if (USF4 && !(sd->harmonic_sbr & XAAC_ESBR_NO_X_DELAY)) rc = -1; /* 4:1 with a transposer's delay: not built */
  /* sbr_qmf_out: 8 rows of history; the stages write every cell of rows 8..31 that is read later, so those are cleared
     only for a frame without SBR processing (the reference zeroes the whole buffer then, sbr_dec.c:956-961) */
  {
#pragma unroll
    for (int r = 0; r < HIST; r++) {
      ore[64 * r + lane] = apply ? hist_re[r] : 0.0f;
      oim[64 * r + lane] = apply ? hist_im[r] : 0.0f;
    }
    /* rows 32.. become the next frame's history: cleared, the stages fill what they reach */
    for (int i = ((!apply || rc) ? HIST : SLOTS) * 64 + lane; i < OUT_ROWS * 64; i += 64) {
      ore[i] = 0.0f;
      oim[i] = 0.0f;
    }
  }
  __syncthreads();
  /* USAC channels (xaac_esbr.h: XAAC_ESBR_USAC / _NO_X_DELAY): no clearing above the old cross-over band (sbr_dec.c:868); without a
     transposer the analysis rows of THIS frame are rows 8..39 of the buffer the tools read (codec_x_delay 0, sbr_dec.c:819-826) */
  const bool no_x_delay = (sd->harmonic_sbr & XAAC_ESBR_NO_X_DELAY) != 0;
  if constexpr (USF4) { /* rows 0..13: the history; 14..77: the analysis bank's, in place; 78, 79: zero */
#pragma unroll
    for (int r = 0; r < HIST; r++) {
      qre[64 * r + lane] = st->qmf_re[r][lane];
      qim[64 * r + lane] = st->qmf_im[r][lane];
    }
    for (int r = HIST + 64; r < XAAC_ESBR_Q_ROWS_4_1; r++) {
      qre[64 * r + lane] = 0.0f;
      qim[64 * r + lane] = 0.0f;
    }
    if (lane >= 32) /* the bank wrote bands 0..31 of its rows (16 of them zeros); the upper half is the reference's never-written zeros */
      for (int r = HIST; r < HIST + 64; r++) {
        qre[64 * r + lane] = 0.0f;
        qim[64 * r + lane] = 0.0f;
      }
    __syncthreads();
  } else if (no_x_delay) {
    for (int j0 = 0; j0 < 32; j0 += 16) {
      float a[16], b[16];
#pragma unroll
      for (int j = 0; j < 16; j++) {
        a[j] = lane < 32 ? are[64 * (j0 + j) + lane] : 0.0f;
        b[j] = lane < 32 ? aim[64 * (j0 + j) + lane] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 16; j++) {
        st->qmf_re[8 + j0 + j][lane] = a[j];
        st->qmf_im[8 + j0 + j][lane] = b[j];
      }
    }
    __syncthreads();
  }
  if (!(sd->harmonic_sbr & XAAC_ESBR_USAC) && sd->qmf_sb_prev >= 0 && sd->qmf_sb_prev <= 64) xe_hbe_history_clear(cx, st, sd->qmf_sb_prev);
  XE_T(1);
  const XeMat src = {qre + 128, qim + 128}, dst = {ore + 128, oim + 128};
  /* the harmonic transposer's rows (sbr_dec.c:859-868): its launches wrote rows 8..39 of the scratch matrix for this frame
     if the channel has one with usable parameters; rows 0..7 are the previous frame's last rows */
  float *phr = p.ph_re + (size_t)ch * XAAC_ESBR_PH_ROWS * 64, *phi = p.ph_im + (size_t)ch * XAAC_ESBR_PH_ROWS * 64;
  bool have_ph = false;
  if constexpr (HARM) {
    if (p.dft) {
      have_ph = apply && p.dft[ch].last_status == 0;
    } else {
      have_ph = p.hbe && apply && xh_apply_params_ok(p.hbe + ch, sd->pitch_in_bins);
      if (have_ph && !XAAC_HBE_LDS_OK(p.hbe[ch].synth_size, p.hbe_lds_synth_size)) have_ph = false, rc = -1; /* the host's hint was wrong */
    }
  }
  if (HARM && have_ph) {
    float t0[8], t1[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      t0[r] = st->ph_re[r][lane];
      t1[r] = st->ph_im[r][lane];
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      phr[64 * r + lane] = t0[r];
      phi[64 * r + lane] = t1[r];
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      t0[r] = phr[64 * (32 + r) + lane];
      t1[r] = phi[64 * (32 + r) + lane];
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      st->ph_re[r][lane] = t0[r];
      st->ph_im[r][lane] = t1[r];
    }
    __syncthreads();
  }
  const XeMat ph = {phr + 128, phi + 128};
  const xaac_esbr_pvc_side *pvs = PVC ? p.pvc_side + ch : nullptr;
  xaac_esbr_pvc_state *pst = PVC ? p.pvc_state + ch : nullptr;
  float *penv = PVC ? p.pvc_out + (size_t)ch * XAAC_PVC_SLOTS * 64 : nullptr;
  if (apply && rc == 0) {
    xe_generate_hf<HARM>(cx, h, f, sd, st, &w, src, dst, have_ph ? &ph : nullptr, RATE);
    __syncthreads();
    XE_T(2);
    if constexpr (PVC) { /* sbr_dec.c:931-953: the low band's energies through the PVC decoder, or its "no PVC frame" bookkeeping */
      if (pvs->sbr_mode == XAAC_ESBR_SBR_PVC) {
        __shared__ XpWork pw;
        const XpCx pcx = {lane, 64};
        if (!w.err && xp_process(pcx, &pw, &pvs->pvc, qre + 128, qim + 128, (size_t)SLOTS * 64, &pst->pvc, penv)) w.err = -1;
      } else if (lane == 0) {
        pst->pvc.prev_pvc_flg = 0;
        pst->pvc.prev_first_bnd_idx = h->sub_band_start;
        pst->pvc.prev_pvc_rate = RATE;
      }
      __syncthreads();
    }
    rc = w.err ? -1 : xe_env_calc(cx, h, f, sd, st, &w, dst, src, (HARM && have_ph) ? (p.dft ? p.dft[ch].x_over_qmf : p.hbe[ch].x_over_qmf) : nullptr, pvs, pst, penv, RATE);
  }
  if (PVC && lane == 0) pst->prev_sbr_mode = pvs->sbr_mode; /* sbr_dec.c:1006 */
  __syncthreads();
  XE_T(3);
  {
    const int stop = apply ? RATE * f->border_vec[0] : 0;
    const int xo_prev = sd->qmf_sb_prev, xo_now = h->sub_band_start;
    for (int i0 = 0; i0 < SLOTS; i0 += 16) { /* regrouping, sbr_dec.c:365-395; sixteen rows (32 words a lane) in flight */
      float a[16], b[16];
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int i = i0 + j, xo = i < stop ? xo_prev : xo_now;
        a[j] = lane < xo ? qre[64 * (2 + i) + lane] : ore[64 * (2 + i) + lane];
        b[j] = lane < xo ? qim[64 * (2 + i) + lane] : oim[64 * (2 + i) + lane];
      }
#pragma unroll
      for (int j = 0; j < 16; j++) {
        rre[64 * (i0 + j) + lane] = a[j];
        rim[64 * (i0 + j) + lane] = b[j];
      }
    }
  }
  if (!USF4 && p.with_ps) /* the six look-ahead rows of the PS hybrid filter: bands 0..4 of qmf_buf rows 34..39 (sbr_dec.c:487-505) */
    for (int i = 32; i < 38; i++) {
      rre[64 * i + lane] = lane < 5 ? st->qmf_re[2 + i][lane] : 0.0f;
      rim[64 * i + lane] = lane < 5 ? st->qmf_im[2 + i][lane] : 0.0f;
    }
  __syncthreads();
  XE_T(4);
  /* histories: rows 32.. of this frame's buffers become rows 0.. of the next frame's (sbr_dec.c:835-857).  Loads of one
     batch all in flight: the two 8-row tails first, then the analysis bank's 32 new rows sixteen at a time */
  if constexpr (USF4) { /* rows 64..77 of both matrices; behind them the state holds the zeros the reference's unwritten rows are */
    float t0[HIST], t1[HIST], u0[HIST], u1[HIST];
#pragma unroll
    for (int r = 0; r < HIST; r++) {
      t0[r] = qre[64 * (64 + r) + lane];
      t1[r] = qim[64 * (64 + r) + lane];
      u0[r] = ore[64 * (64 + r) + lane];
      u1[r] = oim[64 * (64 + r) + lane];
    }
#pragma unroll
    for (int r = 0; r < HIST; r++) {
      st->qmf_re[r][lane] = t0[r];
      st->qmf_im[r][lane] = t1[r];
      if (r < 8) {
        st->out_re[r][lane] = u0[r];
        st->out_im[r][lane] = u1[r];
      } else {
        st->ph_re[r - 8][lane] = u0[r];
        st->ph_im[r - 8][lane] = u1[r];
      }
    }
    for (int r = HIST; r < XAAC_ESBR_HIST_ROWS; r++) {
      st->qmf_re[r][lane] = 0.0f;
      st->qmf_im[r][lane] = 0.0f;
    }
    for (int r = HIST - 8; r < 8; r++) {
      st->ph_re[r][lane] = 0.0f;
      st->ph_im[r][lane] = 0.0f;
    }
  } else {
    float t0[8], t1[8], u0[8], u1[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      t0[r] = st->qmf_re[32 + r][lane];
      t1[r] = st->qmf_im[32 + r][lane];
      u0[r] = ore[64 * (32 + r) + lane];
      u1[r] = oim[64 * (32 + r) + lane];
    }
    float a[16], b[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      a[j] = lane < 32 ? are[64 * j + lane] : 0.0f;
      b[j] = lane < 32 ? aim[64 * j + lane] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      st->qmf_re[r][lane] = t0[r];
      st->qmf_im[r][lane] = t1[r];
      st->out_re[r][lane] = u0[r];
      st->out_im[r][lane] = u1[r];
    }
    /* (codec_x_delay 0: the analysis rows were this frame's rows 8..39 and are not history; the reference's rows 40..71 stay zero) */
#pragma unroll
    for (int j = 0; j < 16; j++) {
      st->qmf_re[8 + j][lane] = no_x_delay ? 0.0f : a[j];
      st->qmf_im[8 + j][lane] = no_x_delay ? 0.0f : b[j];
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
      a[j] = lane < 32 ? are[64 * (16 + j) + lane] : 0.0f;
      b[j] = lane < 32 ? aim[64 * (16 + j) + lane] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
      st->qmf_re[24 + j][lane] = no_x_delay ? 0.0f : a[j];
      st->qmf_im[24 + j][lane] = no_x_delay ? 0.0f : b[j];
    }
  }
#ifdef XE_PROFILE
  XE_T(5);
  if (lane < 16) atomicAdd(reinterpret_cast<unsigned long long *>(p.status) + lane, (unsigned long long)xe_prof_acc[lane]);
#else
  if (lane == 0 && p.status) p.status[ch] = rc;
#endif
}

extern "C" hipError_t xaac_launch_esbr_core(const XaacEsbrCoreParams *p, hipStream_t stream) {
  if (p->usf4) { /* 4:1 SBR: USAC channels without a transposer */
    if (p->pvc_side) hipLaunchKernelGGL((xaac_esbr_core_kernel<false, true, true>), dim3(p->n_ch), dim3(64), 0, stream, *p);
    else hipLaunchKernelGGL((xaac_esbr_core_kernel<false, false, true>), dim3(p->n_ch), dim3(64), 0, stream, *p);
  } else if (p->pvc_side) { /* USAC channels with PVC frames: their own instantiations, the others carry none of that code */
    if (p->hbe || p->dft) hipLaunchKernelGGL((xaac_esbr_core_kernel<true, true>), dim3(p->n_ch), dim3(64), 0, stream, *p);
    else hipLaunchKernelGGL((xaac_esbr_core_kernel<false, true>), dim3(p->n_ch), dim3(64), 0, stream, *p);
  } else if (p->hbe || p->dft) hipLaunchKernelGGL((xaac_esbr_core_kernel<true, false>), dim3(p->n_ch), dim3(64), 0, stream, *p);
  else hipLaunchKernelGGL((xaac_esbr_core_kernel<false, false>), dim3(p->n_ch), dim3(64), 0, stream, *p);
  return hipGetLastError();
}

/* xaac_warm_up (xaac_abi.cpp): asking for a kernel's attributes puts this translation unit's code object on the device */
extern "C" hipError_t xaac_warm_esbr_core(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&xaac_esbr_core_kernel<false, false>));
}
