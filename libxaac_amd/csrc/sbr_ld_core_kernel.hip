/*
 * sbr_ld_core_kernel.hip -- the middle of ixheaacd_sbr_dec for AAC-ELD channels (low-delay SBR: decoder/ixheaacd_sbr_dec.c:
 * 726-775, :1050-1245, :1283-1308 with AOT_ER_AAC_ELD, HQ): block floating point, HF generator, envelope adjuster, state
 * update between the two LD complex banks (sbr_qmf_kernel.hip), on the frame's own 16 or 15 QMF slots -- no overlap slots,
 * one slot per time slot.  Arithmetic: sbr_core.h with the low-delay grid (XsQmfT<1, 64, 1>), the same source the oracle
 * runs (oracle/oracle_sbr.cpp: xo_sbr_dec_eld), pinned on the compiled reference (tests/test_sbr_eld_chains.py).
 *
 * Mapping: one wave = one channel-frame; its 18 x 128-word matrix (two LPC history rows + the slots, 64 real | 64
 * imaginary), header, the head of the frame side info, the state's tail and the scratch in LDS (14.7 KB: ten waves a CU).
 * The frame is a fifth of an HE-AAC one (16 slots, no parametric stereo), so the generic one-envelope chain of sbr_core.h
 * runs as it stands; the tables are read where they are (global memory: a few lookups per envelope).
 */
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define XS_SYNC_WAVE_LDS 1 /* XsCx::sync(): wave-level, LDS only (sbr_core.h) */
#include "sbr_core.h"
#include "sbr_ld_core_kernel.h"

namespace {
struct XlLdsState { /* the state members the core touches: the three bank limits + the struct tail */
  int16_t codec_usb, syn_lsb, syn_usb, pad2_;
  XAAC_SBR_STATE_TAIL_FIELDS
};
constexpr int kHeadOff = offsetof(xaac_sbr_eld_state, codec_usb);
constexpr int kTailOff = offsetof(xaac_sbr_eld_state, lpc_real);
constexpr int kTailWords = (sizeof(xaac_sbr_eld_state) - kTailOff) / 4;
static_assert(kHeadOff % 4 == 0 && kTailOff % 4 == 0 && sizeof(xaac_sbr_eld_state) % 4 == 0 && kTailOff == kHeadOff + 8, "word copies");
static_assert(offsetof(XlLdsState, lpc_real) == 8 && sizeof(XlLdsState) == 8 + kTailWords * 4, "mirror layout");
constexpr int kFrameHeadBytes = offsetof(xaac_sbr_frame, int_env_sf_arr);
static_assert(kFrameHeadBytes % 4 == 0 && offsetof(xaac_sbr_frame, int_noise_floor) % 4 == 0 && sizeof(xaac_sbr_header) % 4 == 0, "word copies");

struct XlLds {
  int32_t x[(2 + 16) * 128];
  XlLdsState st;
  xaac_sbr_header h;
  int32_t f_head[kFrameHeadBytes / 4];
  int32_t noise_floor[sizeof(((xaac_sbr_frame *)0)->int_noise_floor) / 4];
  XsWork w;
};
}  // namespace

__global__ __launch_bounds__(64) void xaac_sbr_ld_core_kernel(XaacSbrLdCoreParams p) {
  __shared__ XlLds s;
  const int ch = blockIdx.x, lane = threadIdx.x, n = p.n_slots;
  xaac_sbr_eld_state *gst = p.state + ch;
  int32_t *gw = reinterpret_cast<int32_t *>(gst);
  int32_t *gx = p.x + (size_t)ch * n * 128; /* the banks' layout: [n_ch][n_slots][128] */
  /* ---- copy-in: side info, state, the analysed slots (bands 0..31 real | imaginary; the rest of a row is cleared below) */
  {
    const int32_t *gh = reinterpret_cast<const int32_t *>(p.header + ch), *gf = reinterpret_cast<const int32_t *>(p.frame + ch);
    for (int i = lane; i < (int)(sizeof(xaac_sbr_header) / 4); i += 64) reinterpret_cast<int32_t *>(&s.h)[i] = gh[i];
    for (int i = lane; i < kFrameHeadBytes / 4; i += 64) s.f_head[i] = gf[i];
    for (int i = lane; i < (int)(sizeof(s.noise_floor) / 4); i += 64)
      s.noise_floor[i] = reinterpret_cast<const int32_t *>(p.frame[ch].int_noise_floor)[i];
    int32_t *m = reinterpret_cast<int32_t *>(&s.st);
    for (int i = lane; i < 2 + kTailWords; i += 64) m[i] = gw[kHeadOff / 4 + i];
    for (int i = lane; i < 2 * 128; i += 64) s.x[i] = 0;
    for (int r = 0; r < n; r++) {
      s.x[(2 + r) * 128 + lane] = (lane < 32) ? gx[r * 128 + lane] : 0;
      s.x[(2 + r) * 128 + 64 + lane] = (lane < 32) ? gx[r * 128 + 64 + lane] : 0;
    }
  }
  __syncthreads();
  const xaac_sbr_frame *f = reinterpret_cast<const xaac_sbr_frame *>(s.f_head); /* head members only */
  const XsCx cx = {lane, 64};
  const XsQmfT<1, 64, 1> x = {s.x};
  int rc = -1, save_lb_scale = 0;
  if (lane == 0) s.st.lb_scale = 0; /* sbr_dec.c:767 */
  const int refused = xs_side_info_bad(cx, &s.h, f, &s.st, 1) || s.h.num_time_slots != n;
  if (!refused) {
    /* (ixheaacd_rescale_x_overlap runs in front of the analysis bank in the reference; the bank took the band limit it leaves
       from the frame, sbr_qmf_kernel.hip, and its clearing of rows does not survive the bank: sbr_core.h) */
    if (f->apply_processing) xs_rescale_x_overlap(cx, &s.h, f, &s.st, x);
    cx.sync();
    if (lane == 0) {
      s.st.st_lb_scale = 0;
      s.st.lb_scale = -9; /* what ixheaacd_cplx_anal_qmffilt leaves for AOT_ER_AAC_ELD (generic:630-636) */
    }
    cx.sync();
    rc = xs_sbr_core(cx, &s.h, f, p.frame[ch].int_env_sf_arr, reinterpret_cast<const int16_t *>(s.noise_floor), &s.st, x, &s.w,
                     static_cast<const int16_t *>(nullptr), &save_lb_scale);
  }
  cx.sync();
  /* ---- copy-out: the synthesis bank's parameters, the matrix, the state */
  if (lane == 0) {
    int16_t *par = p.syn_par + 8 * (size_t)ch;
    par[0] = s.st.lb_scale;
    par[1] = s.st.ov_lb_scale;
    par[2] = s.st.hb_scale;
    par[3] = s.st.st_syn_scale;
    par[4] = s.st.syn_lsb;
    par[5] = s.st.syn_usb;
    par[6] = (int16_t)(rc != 0); /* a refused / failed frame is not synthesised (the reference returns in front of its bank) */
    par[7] = 0;
    if (rc == 0) s.st.ov_lb_scale = (int16_t)save_lb_scale; /* sbr_dec.c:1304 */
    if (p.status) p.status[ch] = rc;
  }
  cx.sync();
  if (rc == 0) {
    for (int r = 0; r < n; r++) {
      gx[r * 128 + lane] = s.x[(2 + r) * 128 + lane];
      gx[r * 128 + 64 + lane] = s.x[(2 + r) * 128 + 64 + lane];
    }
    const int32_t *m = reinterpret_cast<const int32_t *>(&s.st);
    for (int i = lane; i < 2 + kTailWords; i += 64) gw[kHeadOff / 4 + i] = m[i];
  }
}

extern "C" hipError_t xaac_launch_sbr_ld_core(const XaacSbrLdCoreParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_sbr_ld_core_kernel, dim3(p->n_ch), dim3(64), 0, stream, *p);
  return hipGetLastError();
}

/* xaac_warm_up (xaac_abi.cpp) */
extern "C" hipError_t xaac_warm_sbr_ld_core(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&xaac_sbr_ld_core_kernel));
}
