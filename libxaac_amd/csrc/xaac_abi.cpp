/*
 * xaac_abi.cpp -- the C-ABI layer of libxaac_amd.so (include/xaac_amd.h):
 * argument validation, error codes in the reference's IA_ERRORCODE convention,
 * launch geometry, and the host-buffer convenience path.  No arithmetic lives
 * here; the work is in imdct_kernel.hip.  There is deliberately NO CPU fallback:
 * without a HIP device xaac_create fails with XAAC_FATAL_NO_DEVICE.
 */
#include <hip/hip_runtime_api.h>
#include "build_id.h" /* build/build_id.h: XAAC_BUILD_ID (Makefile) */

#include <cstdint>
#include <cstdlib>
#include <new>

#include "../../include/xaac_amd.h"
#include "imdct_kernel.h"
#include "sbr_qmf_kernel.h"
#include "sbr_core_kernel.h"
#include "sbr_ld_core_kernel.h"
#include "sbr_ps_kernel.h"
#include "limiter_kernel.h"
#include "esbr_qmf_kernel.h"
#include "usac_imdct_kernel.h"
#include "imdct960_kernel.h"
#include "imdct_ld_kernel.h"
#include "esbr_core_kernel.h"
#include "hbe_kernel.h"
#include "pvc_kernel.h"
#include <cmath>
#include <cstddef>
#include <cstring>

struct xaac_ctx {
  int device;
  hipStream_t stream;
  bool owns_stream;
  int num_cu;
  int blocks_per_cu;
  int qmf_blocks_per_cu[4];
  int last_grid, last_block, last_lds;
};

/* Profiling builds (-DXS_PROFILE / -DXL_PROFILE, tools/prof_sbr_core.py, tools/time_limiter.py) collect 64-bit phase
   counters in the caller's status buffer; it then has to hold 128 of them, which the tools' batches (>= 256
   channels) do.  Regular builds never hand the kernels a counter buffer. */
#if defined(XS_PROFILE) || defined(XL_PROFILE)
#define XAAC_DBG_BUF(status, n) ((n) >= 256 ? (status) : nullptr)
#else
#define XAAC_DBG_BUF(status, n) static_cast<int32_t *>(nullptr)
#endif

namespace {

inline bool hip_ok(hipError_t e) { return e == hipSuccess; }

/* persistent grid: enough workgroups to fill every CU at the kernel's
   occupancy, never more than the batch needs */
int pick_grid(const xaac_ctx *c, int n_ch) {
  int blocks_needed = (n_ch + XAAC_IMDCT_WAVES - 1) / XAAC_IMDCT_WAVES;
  int resident = c->num_cu * c->blocks_per_cu;
  int g = blocks_needed < resident ? blocks_needed : resident;
  return g < 1 ? 1 : g;
}

int32_t check_batch(const xaac_imdct_batch *b) {
  if (!b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0) return XAAC_FATAL_BAD_ARG;
  if (b->ch_fac != 1 && b->ch_fac != 2) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch % b->ch_fac) return XAAC_FATAL_BAD_ARG;
  if (b->pcm_mode != XAAC_PCM_LC && b->pcm_mode != XAAC_PCM_SBR) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch > 0 && (!b->spec || !b->ics || !b->overlap || !b->state)) return XAAC_FATAL_NULL_ARG;
  return XAAC_OK;
}

}  // namespace

extern "C" {

/* "... build <id>": the sha256 of the library's sources (Makefile: BUILD_ID), so that a counter profile names its library */
const char *xaac_version(void) { return "libxaac_amd 0.4 gfx950 build " XAAC_BUILD_ID; }

int32_t xaac_create(xaac_ctx **out, int32_t device, void *hip_stream) {
  if (!out) return XAAC_FATAL_NULL_ARG;
  *out = nullptr;
  int n = 0;
  if (!hip_ok(hipGetDeviceCount(&n)) || n <= 0) return XAAC_FATAL_NO_DEVICE;
  if (device < 0 || device >= n) return XAAC_FATAL_BAD_ARG;
  if (!hip_ok(hipSetDevice(device))) return XAAC_FATAL_HIP;
  xaac_ctx *c = new (std::nothrow) xaac_ctx();
  if (!c) return XAAC_FATAL_HIP;
  c->device = device;
  c->owns_stream = false;
  c->stream = static_cast<hipStream_t>(hip_stream);
  if (!hip_stream) {
    if (!hip_ok(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking))) {
      delete c;
      return XAAC_FATAL_HIP;
    }
    c->owns_stream = true;
  }
  hipDeviceProp_t prop;
  if (!hip_ok(hipGetDeviceProperties(&prop, device))) {
    if (c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return XAAC_FATAL_HIP;
  }
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->blocks_per_cu = xaac_imdct_blocks_per_cu();
  for (int i = 0; i < 4; i++) c->qmf_blocks_per_cu[i] = xaac_qmf_blocks_per_cu(i);
  c->last_grid = c->last_block = c->last_lds = 0;
  *out = c;
  return XAAC_OK;
}

int32_t xaac_destroy(xaac_ctx *c) {
  if (!c) return XAAC_FATAL_NULL_ARG;
  if (c->owns_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return XAAC_OK;
}

int32_t xaac_set_stream(xaac_ctx *c, void *hip_stream) {
  if (!c) return XAAC_FATAL_NULL_ARG;
  if (c->owns_stream) (void)hipStreamDestroy(c->stream);
  c->owns_stream = false;
  c->stream = static_cast<hipStream_t>(hip_stream);
  return XAAC_OK;
}

extern "C" {
hipError_t xaac_warm_imdct(void), xaac_warm_sbr_qmf(void), xaac_warm_sbr_core(void), xaac_warm_sbr_ps(void), xaac_warm_limiter(void),
    xaac_warm_esbr_qmf(void), xaac_warm_esbr_core(void), xaac_warm_esbr_ps(void), xaac_warm_hbe(void), xaac_warm_usac_imdct(void),
    xaac_warm_imdct960(void), xaac_warm_imdct_ld(void), xaac_warm_pvc(void), xaac_warm_sbr_ld_core(void);
}

int32_t xaac_warm_up(xaac_ctx *c) {
  if (!c) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  hipError_t (*const hooks[])(void) = {xaac_warm_imdct,     xaac_warm_sbr_qmf,   xaac_warm_sbr_core, xaac_warm_sbr_ps,     xaac_warm_limiter,
                                       xaac_warm_esbr_qmf,  xaac_warm_esbr_core, xaac_warm_esbr_ps,  xaac_warm_hbe,        xaac_warm_usac_imdct,
                                       xaac_warm_imdct960,  xaac_warm_imdct_ld,  xaac_warm_pvc,      xaac_warm_sbr_ld_core};
  for (auto h : hooks)
    if (!hip_ok(h())) return XAAC_FATAL_HIP;
  return XAAC_OK;
}

int32_t xaac_sync(xaac_ctx *c) {
  if (!c) return XAAC_FATAL_NULL_ARG;
  return hip_ok(hipStreamSynchronize(c->stream)) ? XAAC_OK : XAAC_FATAL_HIP;
}

int32_t xaac_imdct_process_batch(xaac_ctx *c, const xaac_imdct_batch *b) {
  if (!c) return XAAC_FATAL_NULL_ARG;
  int32_t rc = check_batch(b);
  if (rc != XAAC_OK) return rc;
  if (b->n_ch == 0) return XAAC_OK;
  XaacImdctParams p;
  p.n_ch = b->n_ch;
  p.ch_fac = b->ch_fac;
  p.spec = b->spec;
  p.ics = b->ics;
  p.overlap = b->overlap;
  p.state = b->state;
  p.out32 = b->out32;
  p.pcm16 = b->pcm16;
  p.qshift_adj = b->qshift_adj;
  p.pcm_mode = b->pcm_mode;
  p.status = b->status;
  int grid = pick_grid(c, b->n_ch);
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  if (!hip_ok(xaac_launch_imdct(&p, grid, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = grid;
  c->last_block = XAAC_IMDCT_BLOCK;
  c->last_lds = XAAC_IMDCT_LDS_BYTES;
  return XAAC_OK;
}

int32_t xaac_imdct960_process_batch(xaac_ctx *c, const xaac_imdct_batch *b) {
  if (!c) return XAAC_FATAL_NULL_ARG;
  int32_t rc = check_batch(b);
  if (rc != XAAC_OK) return rc;
  /* the core -> SBR hand-off of a stereo element converts in place, channel after channel (api.c:353-366): channel 1's
     first samples take their low halves from channel 0's PCM.  The 1024-line kernel reproduces that; this one converts
     sample by sample, so the combination is refused rather than answered with different low bits */
  if (b->pcm16 && b->pcm_mode == XAAC_PCM_SBR && b->ch_fac == 2) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  if (!hip_ok(xaac_launch_imdct960(b, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = (b->n_ch + XAAC_I960_WAVES_PER_WG - 1) / XAAC_I960_WAVES_PER_WG;
  c->last_block = 64 * XAAC_I960_WAVES_PER_WG;
  c->last_lds = XAAC_I960_LDS;
  return XAAC_OK;
}

int32_t xaac_imdct_ld_process_batch(xaac_ctx *c, const xaac_imdct_ld_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0 || (b->ch_fac != 1 && b->ch_fac != 2) || b->n_ch % b->ch_fac) return XAAC_FATAL_BAD_ARG;
  if ((b->frame_length != 512 && b->frame_length != 480) || (b->eld != 0 && b->eld != 1)) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->spec || !b->window_shape || !b->overlap || !b->shape_prev || !b->pcm16) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  if (!hip_ok(xaac_launch_imdct_ld(b, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = (b->n_ch + XAAC_LD_WAVES_PER_WG - 1) / XAAC_LD_WAVES_PER_WG;
  c->last_block = 64 * XAAC_LD_WAVES_PER_WG;
  c->last_lds = XAAC_LD_LDS(b->frame_length, b->eld);
  return XAAC_OK;
}

int32_t xaac_imdct_process_batch_host(xaac_ctx *c, const xaac_imdct_batch *hb) {
  if (!c) return XAAC_FATAL_NULL_ARG;
  int32_t rc = check_batch(hb);
  if (rc != XAAC_OK) return rc;
  const size_t n = (size_t)hb->n_ch;
  if (n == 0) return XAAC_OK;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  const size_t sz_spec = n * 1024 * 4, sz_ics = n * sizeof(xaac_ics_info), sz_ovl = n * 512 * 4,
               sz_st = n * sizeof(xaac_ovl_state), sz_o32 = hb->out32 ? n * 1024 * 4 : 0,
               sz_pcm = hb->pcm16 ? n * 1024 * 2 : 0, sz_q = hb->qshift_adj ? n : 0, sz_stat = hb->status ? n * 4 : 0;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t total = up(sz_spec) + up(sz_ics) + up(sz_ovl) + up(sz_st) + up(sz_o32) + up(sz_pcm) + up(sz_q) + up(sz_stat);
  char *d = nullptr;
  if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d), total))) return XAAC_FATAL_HIP;
  char *cur = d;
  auto carve = [&](size_t v) {
    char *r = cur;
    cur += up(v);
    return r;
  };
  xaac_imdct_batch db = *hb;
  char *d_spec = carve(sz_spec), *d_ics = carve(sz_ics), *d_ovl = carve(sz_ovl), *d_st = carve(sz_st);
  char *d_o32 = carve(sz_o32), *d_pcm = carve(sz_pcm), *d_q = carve(sz_q), *d_stat = carve(sz_stat);
  db.spec = reinterpret_cast<const int32_t *>(d_spec);
  db.ics = reinterpret_cast<const xaac_ics_info *>(d_ics);
  db.overlap = reinterpret_cast<int32_t *>(d_ovl);
  db.state = reinterpret_cast<xaac_ovl_state *>(d_st);
  db.out32 = hb->out32 ? reinterpret_cast<int32_t *>(d_o32) : nullptr;
  db.pcm16 = hb->pcm16 ? reinterpret_cast<int16_t *>(d_pcm) : nullptr;
  db.qshift_adj = hb->qshift_adj ? reinterpret_cast<int8_t *>(d_q) : nullptr;
  db.status = hb->status ? reinterpret_cast<int32_t *>(d_stat) : nullptr;
  bool ok = hip_ok(hipMemcpyAsync(d_spec, hb->spec, sz_spec, hipMemcpyHostToDevice, c->stream)) &&
            hip_ok(hipMemcpyAsync(d_ics, hb->ics, sz_ics, hipMemcpyHostToDevice, c->stream)) &&
            hip_ok(hipMemcpyAsync(d_ovl, hb->overlap, sz_ovl, hipMemcpyHostToDevice, c->stream)) &&
            hip_ok(hipMemcpyAsync(d_st, hb->state, sz_st, hipMemcpyHostToDevice, c->stream));
  if (ok) {
    rc = xaac_imdct_process_batch(c, &db);
    ok = (rc == XAAC_OK);
  }
  if (ok) {
    ok = hip_ok(hipMemcpyAsync(hb->overlap, d_ovl, sz_ovl, hipMemcpyDeviceToHost, c->stream)) &&
         hip_ok(hipMemcpyAsync(hb->state, d_st, sz_st, hipMemcpyDeviceToHost, c->stream));
    if (ok && sz_o32) ok = hip_ok(hipMemcpyAsync(hb->out32, d_o32, sz_o32, hipMemcpyDeviceToHost, c->stream));
    if (ok && sz_pcm) ok = hip_ok(hipMemcpyAsync(hb->pcm16, d_pcm, sz_pcm, hipMemcpyDeviceToHost, c->stream));
    if (ok && sz_q) ok = hip_ok(hipMemcpyAsync(hb->qshift_adj, d_q, sz_q, hipMemcpyDeviceToHost, c->stream));
    if (ok && sz_stat) ok = hip_ok(hipMemcpyAsync(hb->status, d_stat, sz_stat, hipMemcpyDeviceToHost, c->stream));
  }
  bool synced = hip_ok(hipStreamSynchronize(c->stream));
  (void)hipFree(d);
  if (rc != XAAC_OK) return rc;
  return (ok && synced) ? XAAC_OK : XAAC_FATAL_HIP;
}

static int qmf_grid(const xaac_ctx *c, int n_ch, int which) {
  int pairs = (n_ch + 1) / 2;
  int need = (pairs + XAAC_QMF_WAVES - 1) / XAAC_QMF_WAVES;
  int resident = c->num_cu * c->qmf_blocks_per_cu[which];
  int g = need < resident ? need : resident;
  return g < 1 ? 1 : g;
}

int32_t xaac_qmf_analysis_batch(xaac_ctx *c, const xaac_qmf_ana_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0 || (b->ch_fac != 1 && b->ch_fac != 2) || b->n_ch % b->ch_fac) return XAAC_FATAL_BAD_ARG;
  if (b->slot_stride < (b->low_pow ? 32 : 96) || b->usb < 0 || b->usb > 32) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->pcm || !b->state || !b->qmf) return XAAC_FATAL_NULL_ARG;
  XaacQmfAnaParams p = {};
  p.n_ch = b->n_ch; p.ch_fac = b->ch_fac; p.low_pow = b->low_pow ? 1 : 0; p.usb = b->usb;
  p.slot_stride = b->slot_stride; p.pcm = b->pcm; p.state = b->state; p.qmf = b->qmf;
  p.state_stride = (int32_t)sizeof(xaac_qmf_ana_state); p.qmf_ch_stride = 32 * b->slot_stride;
  const int grid = qmf_grid(c, b->n_ch, p.low_pow ? 0 : 1);
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  if (!hip_ok(xaac_launch_qmf_analysis(&p, grid, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = grid; c->last_block = XAAC_QMF_BLOCK; c->last_lds = XAAC_QMF_WAVES * XAAC_QMF_ANA_LDS_PER_WAVE;
  return XAAC_OK;
}

int32_t xaac_qmf_analysis_eld_batch(xaac_ctx *c, const xaac_qmf_ana_eld_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0 || (b->n_slots != 16 && b->n_slots != 15) || b->slot_stride < 96 || b->usb < 0 || b->usb > 32) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->pcm || !b->state || !b->qmf) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  if (!hip_ok(xaac_launch_qmf_analysis_eld(b, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = (b->n_ch + 3) / 4; c->last_block = 64; c->last_lds = XAAC_QMF_ELD_LDS;
  return XAAC_OK;
}

int32_t xaac_qmf_synthesis_eld_batch(xaac_ctx *c, const xaac_qmf_syn_eld_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0 || (b->n_slots != 16 && b->n_slots != 15) || b->slot_stride < 128) return XAAC_FATAL_BAD_ARG;
  if (b->lsb < 0 || b->usb < b->lsb || b->usb > 64 || b->split < 0 || b->split > b->n_slots) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->qmf || !b->scale || !b->state || !b->pcm) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  if (!hip_ok(xaac_launch_qmf_synthesis_eld(b, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = (b->n_ch + 3) / 4; c->last_block = 64; c->last_lds = XAAC_QMF_ELD_SYN_LDS;
  return XAAC_OK;
}

int32_t xaac_qmf_synthesis_batch(xaac_ctx *c, const xaac_qmf_syn_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0 || (b->ch_fac != 1 && b->ch_fac != 2) || b->n_ch % b->ch_fac) return XAAC_FATAL_BAD_ARG;
  if (b->slot_stride < (b->low_pow ? 64 : 128)) return XAAC_FATAL_BAD_ARG;
  if (b->lsb < 0 || b->usb < b->lsb || b->usb > 64 || b->split < 0 || b->split > 32) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->qmf || !b->scale || !b->state || !b->pcm) return XAAC_FATAL_NULL_ARG;
  XaacQmfSynParams p = {};
  p.n_ch = b->n_ch; p.ch_fac = b->ch_fac; p.low_pow = b->low_pow ? 1 : 0; p.lsb = b->lsb; p.usb = b->usb;
  p.down_sample = b->down_sample ? 1 : 0;
  p.split = b->split; p.slot_stride = b->slot_stride; p.qmf = b->qmf; p.scale = b->scale; p.state = b->state;
  p.pcm = b->pcm;
  p.state_stride = (int32_t)sizeof(xaac_qmf_syn_state); p.qmf_ch_stride = 32 * b->slot_stride;
  p.scale_stride = 4; p.per_ch_bands = 0;
  const int grid = qmf_grid(c, b->n_ch, p.low_pow ? 2 : 3);
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  if (!hip_ok(xaac_launch_qmf_synthesis(&p, grid, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = grid; c->last_block = XAAC_QMF_BLOCK;
  c->last_lds = XAAC_QMF_WAVES * (p.low_pow ? XAAC_QMF_SYN_LDS_PER_WAVE_LP : XAAC_QMF_SYN_LDS_PER_WAVE_HQ);
  return XAAC_OK;
}

uint64_t xaac_esbr_workspace_bytes(int32_t n_ch) { return n_ch > 0 ? (uint64_t)n_ch * XAAC_ESBR_WS_FLOATS * sizeof(float) : 0; }
uint64_t xaac_esbr_workspace_bytes_ratio(int32_t n_ch, int32_t sbr_ratio) {
  if (sbr_ratio != XAAC_ESBR_RATIO_4_1) return xaac_esbr_workspace_bytes(n_ch);
  return n_ch > 0 ? (uint64_t)n_ch * XAAC_ESBR_WS_FLOATS_4_1 * sizeof(float) : 0;
}


int32_t xaac_esbr_sbr_process_batch(xaac_ctx *c, const xaac_esbr_sbr_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->core || !b->header || !b->frame || !b->side || !b->state || !b->out || !b->workspace) return XAAC_FATAL_NULL_ARG;
  const bool with_ps = b->ps_frame != nullptr;
  if (with_ps != (b->ps_state != nullptr) || with_ps != (b->out_r != nullptr)) return XAAC_FATAL_BAD_ARG;
  if ((b->pvc_side != nullptr) != (b->pvc_state != nullptr)) return XAAC_FATAL_BAD_ARG;
  if (b->sbr_ratio < XAAC_ESBR_RATIO_2_1 || b->sbr_ratio > XAAC_ESBR_RATIO_4_1) return XAAC_FATAL_BAD_ARG;
  if (b->hbe_dft_state) { /* the DFT transposer: in the QMF one's place, 2:1 only */
    if (b->hbe_state || b->sbr_ratio != XAAC_ESBR_RATIO_2_1) return XAAC_FATAL_BAD_ARG;
    if (!b->hbe_dft_cfg_tab || !b->hbe_dft_coef_re || !b->hbe_dft_coef_im) return XAAC_FATAL_NULL_ARG;
  }
  if (b->workspace_bytes < xaac_esbr_workspace_bytes_ratio(b->n_ch, b->sbr_ratio)) return XAAC_FATAL_BAD_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  const size_t n = (size_t)b->n_ch;
  float *ws = static_cast<float *>(b->workspace);
  if (b->sbr_ratio == XAAC_ESBR_RATIO_4_1) { /* 16-channel bank -> 64-slot core -> the synthesis bank in two runs of 32 slots */
    if (with_ps || b->hbe_state) return XAAC_FATAL_BAD_ARG;
    float *q_re = ws, *q_im = q_re + n * XAAC_ESBR_Q_ROWS_4_1 * 64;
    float *o_re = q_im + n * XAAC_ESBR_Q_ROWS_4_1 * 64, *o_im = o_re + n * XAAC_ESBR_OUT_ROWS_4_1 * 64;
    float *s_re = o_im + n * XAAC_ESBR_OUT_ROWS_4_1 * 64, *s_im = s_re + n * 64 * 64;
    float *pvc_o = s_im + n * 64 * 64;
    XaacEsbrAnaNbParams pa = {b->n_ch, 16, 64, b->core, 1024, &b->state->ana, (int32_t)sizeof(xaac_esbr_state),
                              q_re + XAAC_ESBR_OUT_HIST_ROWS_4_1 * 64, q_im + XAAC_ESBR_OUT_HIST_ROWS_4_1 * 64, XAAC_ESBR_Q_ROWS_4_1 * 64};
    if (!hip_ok(xaac_launch_esbr_analysis_nb(&pa, c->stream))) return XAAC_FATAL_HIP;
    XaacEsbrCoreParams pc = {b->n_ch, b->header, b->frame, b->side, b->state, nullptr, nullptr, o_re, o_im, s_re, s_im,
                             0, b->status, nullptr, nullptr, nullptr, 0, b->pvc_side, b->pvc_state, pvc_o, 1, q_re, q_im};
    if (!hip_ok(xaac_launch_esbr_core(&pc, c->stream))) return XAAC_FATAL_HIP;
    for (int half = 0; half < 2; half++) { /* (down-sampled: half the samples of each run, at the same row pitch) */
      XaacEsbrSynParams ps = {b->n_ch, s_re + half * 2048, s_im + half * 2048, &b->state->syn, b->out + half * (b->down_sample ? 1024 : 2048),
                              (int32_t)sizeof(xaac_esbr_state), 64 * 64, nullptr, 4096};
      if (!hip_ok((b->down_sample ? xaac_launch_esbr_synthesis_ds : xaac_launch_esbr_synthesis)(&ps, c->stream))) return XAAC_FATAL_HIP;
    }
    c->last_grid = b->n_ch; c->last_block = 64; c->last_lds = 0;
    return XAAC_OK;
  }
  float *ana_re = ws, *ana_im = ana_re + n * 2048;
  float *out_re = ana_im + n * 2048, *out_im = out_re + n * XAAC_ESBR_OUT_ROWS * 64;
  float *syn_re = out_im + n * XAAC_ESBR_OUT_ROWS * 64, *syn_im = syn_re + n * XAAC_ESBR_L_ROWS * 64;
  float *r_re = syn_im + n * XAAC_ESBR_L_ROWS * 64, *r_im = r_re + n * 2048;
  float *ph_re = r_im + n * 2048, *ph_im = ph_re + n * XAAC_ESBR_PH_ROWS * 64;
  float *pvc_out = ph_im + n * XAAC_ESBR_PH_ROWS * 64;
  /* the banks' states are members of xaac_esbr_state / xaac_esbr_ps_state: the bank kernels take them at that stride */
  if (b->sbr_ratio == XAAC_ESBR_RATIO_8_3) { /* 768 samples a frame through the 24-channel bank; bands 24..31 of the rows zeroed */
    XaacEsbrAnaNbParams pn = {b->n_ch, 24, 32, b->core, 1024, &b->state->ana, (int32_t)sizeof(xaac_esbr_state), ana_re, ana_im, 2048};
    if (!hip_ok(xaac_launch_esbr_analysis_nb(&pn, c->stream))) return XAAC_FATAL_HIP;
  } else {
    XaacEsbrAnaParams pa = {b->n_ch, b->core, &b->state->ana, ana_re, ana_im, (int32_t)sizeof(xaac_esbr_state)};
    if (!hip_ok(xaac_launch_esbr_analysis(&pa, c->stream))) return XAAC_FATAL_HIP;
  }
  if (b->hbe_dft_state) {
    /* sbr_dec.c:880-892 (-esbr_hq:1): the DFT transposer in the QMF one's place -- its output signal, then its analysis bank into
       rows 8..39 of the ph scratch matrix (the reference's clears beyond row 39 left out, sub-bands below a_start zeroed) */
    XaacHbeDftCoreParams dc = {b->n_ch, ana_re, ana_im, nullptr, nullptr, b->hbe_dft_cfg, b->hbe_dft_cfg_tab, b->hbe_dft_state, nullptr, b->frame, b->side};
    if (!hip_ok(xaac_launch_hbe_dft_core(&dc, c->stream))) return XAAC_FATAL_HIP;
    XaacHbeDftParams da = {};
    da.n_ch = b->n_ch; da.no_bins = XAAC_HBE_NO_BINS;
    da.time_in = reinterpret_cast<const float *>(reinterpret_cast<const char *>(b->hbe_dft_state) + offsetof(xaac_hbe_dft_state, output_buf));
    da.in_stride = (int32_t)(sizeof(xaac_hbe_dft_state) / sizeof(float));
    da.coef_re = b->hbe_dft_coef_re; da.coef_im = b->hbe_dft_coef_im; da.cfg = b->hbe_dft_cfg;
    da.state = reinterpret_cast<xaac_hbe_dft_anal_state *>(reinterpret_cast<char *>(b->hbe_dft_state) + offsetof(xaac_hbe_dft_state, anal));
    da.state_stride = (int32_t)sizeof(xaac_hbe_dft_state);
    da.qmf_re = ph_re + 8 * 64; da.qmf_im = ph_im + 8 * 64; da.qmf_stride = XAAC_ESBR_PH_ROWS * 64; da.max_rows = 32; da.zero_below = 1;
    da.chain = 1; da.frame = b->frame;
    da.status_in = reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(b->hbe_dft_state) + offsetof(xaac_hbe_dft_state, last_status));
    da.status_stride = (int32_t)sizeof(xaac_hbe_dft_state);
    if (!hip_ok(xaac_launch_hbe_dft_anal(&da, c->stream))) return XAAC_FATAL_HIP;
  } else if (b->hbe_state) {
    /* sbr_dec.c:882-909: the frame's new analysis rows through the channel's harmonic transposer (two launches),
       its 32 output rows into rows 8..39 of the ph scratch matrix; channels without SBR processing are skipped */
    XaacHbeBanksParams hs = {b->n_ch, XAAC_HBE_NO_BINS, ana_re, ana_im, b->hbe_state, nullptr, nullptr, 1, b->frame, b->side, 2048,
                             XAAC_HBE_PHASE_SYNTH | XAAC_HBE_PHASE_ANAL, b->hbe_max_synth_size};
    if (!hip_ok(xaac_launch_hbe_banks(&hs, c->stream))) return XAAC_FATAL_HIP;
    XaacHbePostParams hp = {b->n_ch, b->hbe_state, nullptr, ph_re + 8 * 64, ph_im + 8 * 64, b->frame, b->side,
                            XAAC_ESBR_PH_ROWS * 64, 1, b->hbe_max_synth_size};
    if (!hip_ok(xaac_launch_hbe_post(&hp, c->stream))) return XAAC_FATAL_HIP;
  }
  XaacEsbrCoreParams pc = {b->n_ch, b->header, b->frame, b->side, b->state, ana_re, ana_im, out_re, out_im, syn_re, syn_im,
                           with_ps ? 1 : 0, b->status, b->hbe_state, ph_re, ph_im, b->hbe_max_synth_size,
                           b->pvc_side, b->pvc_state, pvc_out};
  pc.dft = b->hbe_dft_state;
  if (!hip_ok(xaac_launch_esbr_core(&pc, c->stream))) return XAAC_FATAL_HIP;
  if (with_ps) {
    XaacEsbrPsParams pp = {b->n_ch, b->header, b->frame, b->ps_frame, b->ps_state, syn_re, syn_im, r_re, r_im, b->status};
    if (!hip_ok(xaac_launch_esbr_ps(&pp, c->stream))) return XAAC_FATAL_HIP;
  }
  const auto syn = b->down_sample ? xaac_launch_esbr_synthesis_ds : xaac_launch_esbr_synthesis;
  XaacEsbrSynParams ps = {b->n_ch, syn_re, syn_im, &b->state->syn, b->out, (int32_t)sizeof(xaac_esbr_state), XAAC_ESBR_L_ROWS * 64, nullptr, 2048};
  if (!hip_ok(syn(&ps, c->stream))) return XAAC_FATAL_HIP;
  if (with_ps) {
    XaacEsbrSynParams pr = {b->n_ch, r_re, r_im, &b->ps_state->syn_r, b->out_r, (int32_t)sizeof(xaac_esbr_ps_state), 2048, b->header, 2048};
    if (!hip_ok(syn(&pr, c->stream))) return XAAC_FATAL_HIP;
  }
  c->last_grid = b->n_ch; c->last_block = 64; c->last_lds = 0;
  return XAAC_OK;
}

int32_t xaac_esbr_core_from_pcm16_batch(xaac_ctx *c, const xaac_esbr_core_in_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0 || (b->ch_fac != 1 && b->ch_fac != 2) || b->n_ch % b->ch_fac) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->pcm || !b->core) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  XaacEsbrCoreInParams p = {b->n_ch, b->ch_fac, b->pcm, b->core};
  if (!hip_ok(xaac_launch_esbr_core_from_pcm16(&p, c->stream))) return XAAC_FATAL_HIP;
  return XAAC_OK;
}

int32_t xaac_esbr_pcm16_from_float_batch(xaac_ctx *c, const xaac_esbr_pcm_out_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n < 0 || b->stride < 2048) return XAAC_FATAL_BAD_ARG;
  if (b->n == 0) return XAAC_OK;
  if (!b->left || !b->right || !b->pcm) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  XaacEsbrPcmOutParams p = {b->n, b->stride, b->left, b->right, b->pcm};
  if (!hip_ok(xaac_launch_esbr_pcm16_from_float(&p, c->stream))) return XAAC_FATAL_HIP;
  return XAAC_OK;
}

int32_t xaac_sbr_state_handover(xaac_ctx *c, const xaac_sbr_handover_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n < 0 || (b->mode != XAAC_HANDOVER_PS_START && b->mode != XAAC_HANDOVER_STEREO_START)) return XAAC_FATAL_BAD_ARG;
  if (b->n == 0) return XAAC_OK;
  if (!b->src || !b->dst || !b->state || (b->mode == XAAC_HANDOVER_PS_START && !b->ps_state)) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  if (!hip_ok(xaac_launch_sbr_handover(b, c->stream))) return XAAC_FATAL_HIP;
  return XAAC_OK;
}

int32_t xaac_sbr_state_apply_side_batch(xaac_ctx *c, const xaac_sbr_apply_side_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_streams < 0 || (b->ch_fac != 1 && b->ch_fac != 2)) return XAAC_FATAL_BAD_ARG;
  if (b->n_streams == 0) return XAAC_OK;
  if (!b->header || !b->flags || !b->state) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  if (!hip_ok(xaac_launch_sbr_apply_side(b, c->stream))) return XAAC_FATAL_HIP;
  return XAAC_OK;
}

int32_t xaac_usac_imdct_process_batch(xaac_ctx *c, const xaac_usac_imdct_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (b->ccfl != 0 && b->ccfl != 1024 && b->ccfl != 768) return XAAC_FATAL_BAD_ARG;
  if (!b->coef || !b->ics || !b->overlap || !b->shape_prev) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  const xaac_usac_fac *fac = b->fac;
  if (b->fac_in) { /* the FAC signals of this batch's LPD -> FD transitions, made on the device first */
    if (!b->fac_work || !b->lpd_flags) return XAAC_FATAL_NULL_ARG;
    XaacUsacFacParams pf = {b->n_ch, b->ccfl ? b->ccfl : 1024, b->ics, b->lpd_flags, b->fac_in, b->fac_work};
    if (!hip_ok(xaac_launch_usac_fac(&pf, c->stream))) return XAAC_FATAL_HIP;
    fac = b->fac_work;
  }
  XaacUsacImdctParams p = {b->n_ch, b->ccfl ? b->ccfl : 1024, b->coef, b->ics, b->overlap, b->shape_prev, b->out32, b->time, b->status,
                           b->lpd_flags, fac};
  if (!hip_ok(xaac_launch_usac_imdct(&p, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = (b->n_ch + XAAC_USAC_WAVES_PER_WG - 1) / XAAC_USAC_WAVES_PER_WG;
  c->last_block = 64 * XAAC_USAC_WAVES_PER_WG;
  c->last_lds = XAAC_USAC_LDS;
  return XAAC_OK;
}

int32_t xaac_esbr_qmf_analysis_batch(xaac_ctx *c, const xaac_esbr_ana_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->core || !b->state || !b->qmf_re || !b->qmf_im) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  XaacEsbrAnaParams p = {b->n_ch, b->core, b->state, b->qmf_re, b->qmf_im, (int32_t)sizeof(xaac_esbr_ana_state)};
  if (!hip_ok(xaac_launch_esbr_analysis(&p, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = (b->n_ch + 1) / 2; c->last_block = 64; c->last_lds = XAAC_ESBR_ANA_LDS;
  return XAAC_OK;
}

int32_t xaac_esbr_qmf_analysis_nb_batch(xaac_ctx *c, const xaac_esbr_ana_nb_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0 || (b->n_bands != 24 && b->n_bands != 16) || b->n_slots < 0 || b->n_slots > (b->n_bands == 24 ? 32 : 64) ||
      b->n_bands * b->n_slots > 1024 ||
      b->core_stride < b->n_bands * b->n_slots)
    return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->core || !b->state || !b->qmf_re || !b->qmf_im) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  XaacEsbrAnaNbParams p = {b->n_ch, b->n_bands, b->n_slots, b->core, b->core_stride, b->state, (int32_t)sizeof(xaac_esbr_ana_state),
                           b->qmf_re, b->qmf_im, b->n_slots * 64};
  if (!hip_ok(xaac_launch_esbr_analysis_nb(&p, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = b->n_ch; c->last_block = 64; c->last_lds = 0;
  return XAAC_OK;
}

int32_t xaac_esbr_qmf_synthesis_batch(xaac_ctx *c, const xaac_esbr_syn_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->qmf_re || !b->qmf_im || !b->state || !b->out) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  XaacEsbrSynParams p = {b->n_ch, b->qmf_re, b->qmf_im, b->state, b->out, (int32_t)sizeof(xaac_esbr_syn_state), 2048};
  if (!hip_ok(xaac_launch_esbr_synthesis(&p, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = (b->n_ch + 1) / 2; c->last_block = 64; c->last_lds = XAAC_ESBR_SYN_LDS;
  return XAAC_OK;
}

int32_t xaac_esbr_qmf_synthesis_ds_batch(xaac_ctx *c, const xaac_esbr_syn_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->qmf_re || !b->qmf_im || !b->state || !b->out) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  XaacEsbrSynParams p = {b->n_ch, b->qmf_re, b->qmf_im, b->state, b->out, (int32_t)sizeof(xaac_esbr_syn_state), 2048, nullptr, 1024};
  if (!hip_ok(xaac_launch_esbr_synthesis_ds(&p, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = (b->n_ch + 1) / 2; c->last_block = 64; c->last_lds = 0;
  return XAAC_OK;
}

int32_t xaac_hbe_real_synth_batch(xaac_ctx *c, const xaac_hbe_synth_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0 || b->num_columns < 0 || b->num_columns > XAAC_HBE_NO_BINS) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0 || b->num_columns == 0) return XAAC_OK;
  if (!b->qmf_re || !b->qmf_im || !b->state) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  XaacHbeBanksParams p = {b->n_ch, b->num_columns, b->qmf_re, b->qmf_im, b->state, b->status, nullptr, 0, nullptr, nullptr,
                          b->num_columns * 64, XAAC_HBE_PHASE_SYNTH};
  if (!hip_ok(xaac_launch_hbe_banks(&p, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = b->n_ch; c->last_block = XAAC_HBE_BANKS_THREADS; c->last_lds = XAAC_HBE_BANKS_LDS;
  return XAAC_OK;
}

int32_t xaac_hbe_cplx_anal_batch(xaac_ctx *c, const xaac_hbe_anal_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->state) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  XaacHbeBanksParams p = {b->n_ch, 0, nullptr, nullptr, b->state, b->status, nullptr, 0, nullptr, nullptr, 0, XAAC_HBE_PHASE_ANAL};
  if (!hip_ok(xaac_launch_hbe_banks(&p, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = b->n_ch; c->last_block = XAAC_HBE_BANKS_THREADS; c->last_lds = XAAC_HBE_BANKS_LDS;
  return XAAC_OK;
}

int32_t xaac_hbe_dft_anal_batch_run(xaac_ctx *c, const xaac_hbe_dft_anal_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0 || b->no_bins < 1 || b->no_bins > XAAC_HBE_NO_BINS || b->in_stride < 1) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->time_in || !b->coef_re || !b->coef_im || !b->state || !b->qmf_re || !b->qmf_im) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  XaacHbeDftParams p = {b->n_ch, b->no_bins, b->time_in, b->in_stride, b->coef_re, b->coef_im, b->cfg, b->state, b->qmf_re, b->qmf_im, b->status};
  if (!hip_ok(xaac_launch_hbe_dft_anal(&p, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = b->n_ch; c->last_block = 256; c->last_lds = XAAC_HBE_DFT_LDS;
  return XAAC_OK;
}

int32_t xaac_hbe_dft_apply_batch_run(xaac_ctx *c, const xaac_hbe_dft_apply_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->qmf_re || !b->qmf_im || !b->cfg_tab || !b->coef_re || !b->coef_im || !b->state || !b->pv_re || !b->pv_im || !b->status)
    return XAAC_FATAL_NULL_ARG; /* (status is how the second launch learns which channels the first one refused) */
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  XaacHbeDftCoreParams pc = {b->n_ch, b->qmf_re, b->qmf_im, b->pitch_in_bins, b->oversampling, b->cfg, b->cfg_tab, b->state, b->status, nullptr, nullptr};
  if (!hip_ok(xaac_launch_hbe_dft_core(&pc, c->stream))) return XAAC_FATAL_HIP;
  XaacHbeDftParams pa = {};
  pa.n_ch = b->n_ch; pa.no_bins = XAAC_HBE_NO_BINS;
  pa.time_in = reinterpret_cast<const float *>(reinterpret_cast<const char *>(b->state) + offsetof(xaac_hbe_dft_state, output_buf));
  pa.in_stride = (int32_t)(sizeof(xaac_hbe_dft_state) / sizeof(float));
  pa.coef_re = b->coef_re; pa.coef_im = b->coef_im; pa.cfg = b->cfg;
  pa.state = reinterpret_cast<xaac_hbe_dft_anal_state *>(reinterpret_cast<char *>(b->state) + offsetof(xaac_hbe_dft_state, anal));
  pa.state_stride = (int32_t)sizeof(xaac_hbe_dft_state);
  pa.qmf_re = b->pv_re; pa.qmf_im = b->pv_im; pa.status = b->status; pa.chain = 1;
  pa.status_in = b->status; pa.status_stride = 4;
  if (b->rows32) pa.qmf_stride = 32 * 64, pa.max_rows = 32, pa.zero_below = 1;
  if (!hip_ok(xaac_launch_hbe_dft_anal(&pa, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = b->n_ch; c->last_block = XAAC_HBE_DFT_CORE_THREADS; c->last_lds = XAAC_HBE_DFT_CORE_LDS;
  return XAAC_OK;
}

int32_t xaac_pvc_process_batch(xaac_ctx *c, const xaac_pvc_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0 || b->qmf_stride < 32 * 64) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->frame || !b->qmf_re || !b->qmf_im || !b->state || !b->out) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  XaacPvcParams p = {b->n_ch, b->frame, b->qmf_re, b->qmf_im, b->qmf_stride, b->state, b->out, b->status};
  if (!hip_ok(xaac_launch_pvc(&p, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = b->n_ch; c->last_block = 64; c->last_lds = (int32_t)(sizeof(float) * (31 * 3 + 16 * 3 + 16 * 3 + 16 * 8) + sizeof(xaac_pvc_frame));
  return XAAC_OK;
}

int32_t xaac_hbe_apply_batch(xaac_ctx *c, const xaac_hbe_apply_batch_desc *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->qmf_re || !b->qmf_im || !b->state || !b->pv_re || !b->pv_im) return XAAC_FATAL_NULL_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  /* two launches on the context's stream: the two polyphase banks (with the frame's shift / re-initialisation and the
     parameter check that sets status), then products + output rows */
  XaacHbeBanksParams ps = {b->n_ch, XAAC_HBE_NO_BINS, b->qmf_re, b->qmf_im, b->state, b->status, b->pitch_in_bins, 1, nullptr, nullptr, 2048,
                           XAAC_HBE_PHASE_SYNTH | XAAC_HBE_PHASE_ANAL, b->max_synth_size};
  if (!hip_ok(xaac_launch_hbe_banks(&ps, c->stream))) return XAAC_FATAL_HIP;
  XaacHbePostParams pp = {b->n_ch, b->state, b->pitch_in_bins, b->pv_re, b->pv_im, nullptr, nullptr, 2048, 0, b->max_synth_size};
  if (!hip_ok(xaac_launch_hbe_post(&pp, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = b->n_ch; c->last_block = XAAC_HBE_POST_THREADS; c->last_lds = XAAC_HBE_POST_LDS;
  return XAAC_OK;
}

uint64_t xaac_sbr_lp_workspace_bytes(int32_t n_ch) {
  if (n_ch < 0) return 0;
  return (uint64_t)n_ch * (XAAC_SBR_X_WORDS * 4 + 8 * 2) + 256 + 128; /* matrix, synthesis parameters, the core's counters */
}

int32_t xaac_sbr_lp_process_batch(xaac_ctx *c, const xaac_sbr_lp_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0) return XAAC_FATAL_BAD_ARG;
  if ((b->in_ch_fac != 1 && b->in_ch_fac != 2) || (b->out_ch_fac != 1 && b->out_ch_fac != 2)) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch % b->in_ch_fac || b->n_ch % b->out_ch_fac) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->pcm_in || !b->header || !b->frame || !b->state || !b->pcm_out || !b->workspace) return XAAC_FATAL_NULL_ARG;
  if (b->workspace_bytes < xaac_sbr_lp_workspace_bytes(b->n_ch)) return XAAC_FATAL_BAD_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  int32_t *x = reinterpret_cast<int32_t *>(((uintptr_t)b->workspace + 255) & ~(uintptr_t)255);
  int16_t *par = reinterpret_cast<int16_t *>(x + (size_t)b->n_ch * XAAC_SBR_X_WORDS);
  char *st = reinterpret_cast<char *>(b->state);
  /* 1. analysis bank: 32 new slots into rows 8..39 of each channel's matrix */
  XaacQmfAnaParams pa = {};
  pa.n_ch = b->n_ch; pa.ch_fac = b->in_ch_fac; pa.low_pow = 1; pa.usb = 32; pa.slot_stride = 64;
  pa.state_stride = (int32_t)sizeof(xaac_sbr_state); pa.qmf_ch_stride = XAAC_SBR_X_WORDS;
  pa.pcm = b->pcm_in;
  pa.state = reinterpret_cast<xaac_qmf_ana_state *>(st + offsetof(xaac_sbr_state, ana_ring));
  pa.qmf = x + (2 + 6) * 64;
  /* the core's work counter (and its neighbour) behind the synthesis parameters; the analysis launch clears them on its way */
  int32_t *counters = reinterpret_cast<int32_t *>(((uintptr_t)(par + (size_t)b->n_ch * 8) + 63) & ~(uintptr_t)63);
  pa.zero_words = counters;
  if (!hip_ok(xaac_launch_qmf_analysis(&pa, qmf_grid(c, b->n_ch, 0), c->stream))) return XAAC_FATAL_HIP;
  /* 2. everything between the banks */
  XaacSbrCoreParams pc = {};
  pc.n_ch = b->n_ch; pc.header = b->header; pc.frame = b->frame; pc.state = b->state; pc.x = x; pc.syn_par = par;
  pc.status = b->status;
  pc.defer_count = counters; pc.work_counter = counters + 1; pc.num_cu = c->num_cu; pc.counters_zeroed = 1;
  if (!hip_ok(xaac_launch_sbr_core_lp(&pc, c->stream))) return XAAC_FATAL_HIP;
  /* 3. synthesis bank over rows 2..33 (the 6 delayed + first 26 new slots) */
  XaacQmfSynParams ps = {};
  ps.n_ch = b->n_ch; ps.ch_fac = b->out_ch_fac; ps.low_pow = 1; ps.lsb = 0; ps.usb = 0; ps.split = 6;
  ps.down_sample = b->down_sample ? 1 : 0;
  ps.slot_stride = 64; ps.state_stride = (int32_t)sizeof(xaac_sbr_state); ps.qmf_ch_stride = XAAC_SBR_X_WORDS;
  ps.scale_stride = 8; ps.per_ch_bands = 1;
  ps.qmf = x + 2 * 64; ps.scale = par; ps.dbg = XAAC_DBG_BUF(b->status, b->n_ch);
  ps.state = reinterpret_cast<xaac_qmf_syn_state *>(st + offsetof(xaac_sbr_state, syn_ring));
  ps.pcm = b->pcm_out;
  const int grid = qmf_grid(c, b->n_ch, 2);
  if (!hip_ok(xaac_launch_qmf_synthesis(&ps, grid, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = grid; c->last_block = XAAC_QMF_BLOCK; c->last_lds = XAAC_QMF_WAVES * XAAC_QMF_SYN_LDS_PER_WAVE_LP;
  return XAAC_OK;
}

uint64_t xaac_sbr_hq_workspace_bytes(int32_t n_ch, int32_t with_ps) {
  if (n_ch < 0) return 0;
  uint64_t per = 2 * XAAC_SBR_X_WORDS * 4 + 8 * 2 + 4; /* matrix, synthesis parameters, an entry of the core's stream list */
  if (with_ps) per += 32 * 128 * 4 + 8 * 2;
  return (uint64_t)n_ch * per + 512 + 128;
}

/* the low-delay SBR chain of AAC-ELD channels: LD analysis bank -> core (sbr_ld_core_kernel.hip) -> LD synthesis bank */
uint64_t xaac_sbr_eld_workspace_bytes(int32_t n_ch) {
  return n_ch > 0 ? (uint64_t)n_ch * (16 * 128 * 4 + 8 * 2) + 512 : 0; /* the matrices, the synthesis parameters, alignment */
}
int32_t xaac_sbr_eld_process_batch(xaac_ctx *c, const xaac_sbr_eld_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0 || (b->n_slots != 16 && b->n_slots != 15)) return XAAC_FATAL_BAD_ARG;
  if ((b->in_ch_fac != 1 && b->in_ch_fac != 2) || (b->out_ch_fac != 1 && b->out_ch_fac != 2)) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch % b->in_ch_fac || b->n_ch % b->out_ch_fac) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->pcm_in || !b->header || !b->frame || !b->state || !b->pcm_out || !b->workspace) return XAAC_FATAL_NULL_ARG;
  if (b->workspace_bytes < xaac_sbr_eld_workspace_bytes(b->n_ch)) return XAAC_FATAL_BAD_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  const size_t n = (size_t)b->n_ch;
  int32_t *x = reinterpret_cast<int32_t *>(((uintptr_t)b->workspace + 255) & ~(uintptr_t)255);
  int16_t *par = reinterpret_cast<int16_t *>(x + n * 16 * 128);
  char *st = reinterpret_cast<char *>(b->state);
  XaacQmfEldChain cn = {};
  cn.state_stride = (int32_t)sizeof(xaac_sbr_eld_state);
  /* 1. analysis: n_slots slots of 32 real | 32 imaginary bands into each channel's matrix */
  xaac_qmf_ana_eld_batch a = {};
  a.n_ch = b->n_ch; a.n_slots = b->n_slots; a.usb = 32; a.slot_stride = 128; a.pcm = b->pcm_in;
  a.state = reinterpret_cast<xaac_qmf_ana_eld_state *>(st + offsetof(xaac_sbr_eld_state, ana));
  a.qmf = x; a.status = nullptr;
  cn.pcm_ch_fac = b->in_ch_fac; cn.frame = b->frame;
  cn.codec_usb = reinterpret_cast<const int16_t *>(st + offsetof(xaac_sbr_eld_state, codec_usb));
  if (!hip_ok(xaac_launch_qmf_analysis_eld_chain(&a, &cn, c->stream))) return XAAC_FATAL_HIP;
  /* 2. between the banks */
  XaacSbrLdCoreParams pc = {b->n_ch, b->n_slots, b->header, b->frame, b->state, x, par, b->status};
  if (!hip_ok(xaac_launch_sbr_ld_core(&pc, c->stream))) return XAAC_FATAL_HIP;
  /* 3. synthesis */
  xaac_qmf_syn_eld_batch sy = {};
  sy.n_ch = b->n_ch; sy.n_slots = b->n_slots; sy.split = 0; sy.slot_stride = 128; sy.qmf = x; sy.scale = par;
  sy.state = reinterpret_cast<xaac_qmf_syn_eld_state *>(st + offsetof(xaac_sbr_eld_state, syn));
  sy.pcm = b->pcm_out; sy.status = nullptr; sy.qmf_scaled = b->qmf_handed_on;
  cn.pcm_ch_fac = b->out_ch_fac; cn.syn_par = par;
  if (!hip_ok(xaac_launch_qmf_synthesis_eld_chain(&sy, &cn, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = b->n_ch; c->last_block = 64; c->last_lds = 0;
  return XAAC_OK;
}

int32_t xaac_sbr_hq_process_batch(xaac_ctx *c, const xaac_sbr_hq_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_ch < 0) return XAAC_FATAL_BAD_ARG;
  const bool with_ps = b->ps_frame != nullptr;
  if ((b->ps_frame == nullptr) != (b->ps_state == nullptr)) return XAAC_FATAL_BAD_ARG;
  if (b->in_ch_fac != 1 && b->in_ch_fac != 2) return XAAC_FATAL_BAD_ARG;
  if (!with_ps && b->out_ch_fac != 1 && b->out_ch_fac != 2) return XAAC_FATAL_BAD_ARG;
  if (with_ps && b->down_sample) return XAAC_FATAL_BAD_ARG; /* see xaac_sbr_hq_batch.down_sample */
  if (b->n_ch % b->in_ch_fac || (!with_ps && b->n_ch % b->out_ch_fac)) return XAAC_FATAL_BAD_ARG;
  if (b->n_ch == 0) return XAAC_OK;
  if (!b->pcm_in || !b->header || !b->frame || !b->state || !b->pcm_out || !b->workspace) return XAAC_FATAL_NULL_ARG;
  if (b->workspace_bytes < xaac_sbr_hq_workspace_bytes(b->n_ch, with_ps)) return XAAC_FATAL_BAD_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  const size_t n = (size_t)b->n_ch;
  const int xw = 2 * XAAC_SBR_X_WORDS; /* 40 rows of 64 real | 64 imaginary per stream */
  int32_t *x = reinterpret_cast<int32_t *>(((uintptr_t)b->workspace + 255) & ~(uintptr_t)255);
  int32_t *xr = x + n * xw;
  int16_t *par_l = reinterpret_cast<int16_t *>(xr + (with_ps ? n * 32 * 128 : 0));
  int16_t *par_r = par_l + n * 8;
  char *st = reinterpret_cast<char *>(b->state);
  /* 1. complex analysis bank: 32 new slots into rows 8..39 of each stream's matrix */
  XaacQmfAnaParams pa = {};
  pa.n_ch = b->n_ch; pa.ch_fac = b->in_ch_fac; pa.low_pow = 0; pa.usb = 32; pa.slot_stride = 128;
  pa.state_stride = (int32_t)sizeof(xaac_sbr_state); pa.qmf_ch_stride = xw;
  pa.pcm = b->pcm_in;
  pa.state = reinterpret_cast<xaac_qmf_ana_state *>(st + offsetof(xaac_sbr_state, ana_ring));
  pa.qmf = x + (2 + 6) * 128;
  pa.frame = b->frame;
  /* the core's two counters (streams deferred to the 64-band rows, next stream of the persistent waves) sit behind the
     synthesis parameters; the analysis launch clears them on its way */
  int32_t *counters = reinterpret_cast<int32_t *>(((uintptr_t)(par_l + n * 8 * (with_ps ? 2 : 1)) + 63) & ~(uintptr_t)63);
  pa.zero_words = counters;
  if (!hip_ok(xaac_launch_qmf_analysis(&pa, qmf_grid(c, b->n_ch, 1), c->stream))) return XAAC_FATAL_HIP;
  /* 2. everything between the banks */
  XaacSbrCoreParams pc = {};
  pc.n_ch = b->n_ch; pc.header = b->header; pc.frame = b->frame; pc.state = b->state; pc.x = x; pc.syn_par = par_l;
  pc.status = b->status;
  pc.defer_count = counters; pc.work_counter = counters + 1; pc.defer_list = counters + 2; pc.num_cu = c->num_cu;
  pc.counters_zeroed = 1;
  pc.narrow_only = b->max_band_hint == XAAC_SBR_NARROW_BANDS ? 1 : 0;
  if (!hip_ok(xaac_launch_sbr_core_hq(&pc, c->stream))) return XAAC_FATAL_HIP;
  /* 3. parametric stereo: rows 2..33 become the left channel, xr the right one */
  if (with_ps) {
    XaacPsParams pp;
    pp.n = b->n_ch; pp.x = x; pp.xr = xr; pp.header = b->header; pp.sbr_frame = b->frame; pp.frame = b->ps_frame;
    pp.state = b->ps_state; pp.sbr_state = b->state; pp.par_l = par_l; pp.par_r = par_r; pp.status = b->status; pp.dbg = XAAC_DBG_BUF(b->status, b->n_ch);
    if (!hip_ok(xaac_launch_ps(&pp, c->stream))) return XAAC_FATAL_HIP;
  }
  /* 4. synthesis bank(s) over the 6 delayed + first 26 new slots */
  if (with_ps) { /* both banks of a stream in one wave, interleaved L,R out */
    XaacQmfSynPairParams pq = {};
    pq.n = b->n_ch; pq.split = 6;
    pq.qmf[0] = x + 2 * 128; pq.qmf_stride[0] = xw; pq.scale[0] = par_l;
    pq.state[0] = reinterpret_cast<xaac_qmf_syn_state *>(st + offsetof(xaac_sbr_state, syn_ring));
    pq.state_stride[0] = (int32_t)sizeof(xaac_sbr_state);
    pq.qmf[1] = xr; pq.qmf_stride[1] = 32 * 128; pq.scale[1] = par_r;
    pq.state[1] = reinterpret_cast<xaac_qmf_syn_state *>(reinterpret_cast<char *>(b->ps_state) +
                                                         offsetof(xaac_ps_state, syn_ring_r));
    pq.state_stride[1] = (int32_t)sizeof(xaac_ps_state);
    pq.pcm = b->pcm_out; pq.status = b->status;
    if (!hip_ok(xaac_launch_qmf_synthesis_pair(&pq, c->stream))) return XAAC_FATAL_HIP;
    c->last_grid = b->n_ch; c->last_block = 128; c->last_lds = XAAC_QMF_SYN_PAIR_LDS;
    return XAAC_OK;
  }
  XaacQmfSynParams ps = {};
  ps.n_ch = b->n_ch; ps.ch_fac = b->out_ch_fac; ps.low_pow = 0; ps.split = 6;
  ps.down_sample = b->down_sample ? 1 : 0;
  ps.slot_stride = 128; ps.state_stride = (int32_t)sizeof(xaac_sbr_state); ps.qmf_ch_stride = xw;
  ps.scale_stride = 8; ps.per_ch_bands = 1;
  ps.qmf = x + 2 * 128; ps.scale = par_l; ps.dbg = XAAC_DBG_BUF(b->status, b->n_ch);
  ps.state = reinterpret_cast<xaac_qmf_syn_state *>(st + offsetof(xaac_sbr_state, syn_ring));
  ps.pcm = b->pcm_out;
  const int grid = qmf_grid(c, b->n_ch, 3);
  if (!hip_ok(xaac_launch_qmf_synthesis(&ps, grid, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = grid; c->last_block = XAAC_QMF_BLOCK; c->last_lds = XAAC_QMF_WAVES * XAAC_QMF_SYN_LDS_PER_WAVE_HQ;
  return XAAC_OK;
}

/* ixheaacd_peak_limiter_init, decoder/ixheaacd_peak_limiter.c:46-77 (host side: a stream's state is made once) */
int32_t xaac_peak_limiter_init(xaac_limiter_state *s, uint32_t num_channels, uint32_t sample_rate) {
  if (!s) return XAAC_FATAL_NULL_ARG;
  const uint32_t attack = (uint32_t)(5.0f * sample_rate / 1000);
  if (attack < 1 || attack > XAAC_LIM_MAX_ATTACK || num_channels < 1 || num_channels > XAAC_LIM_MAX_CH)
    return XAAC_FATAL_BAD_ARG;
  std::memset(s, 0, sizeof(*s));
  s->attack_time_samples = attack;
  s->attack_constant = (float)std::pow(0.1, 1.0 / (attack + 1));
  s->release_constant = (float)std::pow(0.1, 1.0 / (50.0f * sample_rate / 1000 + 1));
  s->num_channels = num_channels;
  s->min_gain = 1.0f;
  s->limiter_on = 1;
  s->pre_smoothed_gain = 1.0f;
  s->gain_modified = 1.0f;
  return (int32_t)attack;
}

/* target gains / gains of the streams whose smoothing recursion has to run (4 KB each) + two words of hand-over */
uint64_t xaac_peak_limiter_workspace_bytes(int32_t n_streams) {
  if (n_streams < 0) return 0;
  return (uint64_t)n_streams * (1024 * sizeof(float) + 2 * sizeof(int32_t)) + 256;
}

int32_t xaac_peak_limiter_process_batch(xaac_ctx *c, const xaac_limiter_batch *b) {
  if (!c || !b) return XAAC_FATAL_NULL_ARG;
  if (b->n_streams < 0 || b->frame_len < 1 || b->frame_len > 1024) return XAAC_FATAL_BAD_ARG;
  if (b->num_channels < 1 || b->num_channels > XAAC_LIM_MAX_CH) return XAAC_FATAL_BAD_ARG;
  if (b->stride < (int64_t)b->frame_len * b->num_channels) return XAAC_FATAL_BAD_ARG;
  if (b->n_streams == 0) return XAAC_OK;
  if (!b->samples || !b->qshift_adj || !b->state || !b->workspace) return XAAC_FATAL_NULL_ARG;
  if (b->workspace_bytes < xaac_peak_limiter_workspace_bytes(b->n_streams)) return XAAC_FATAL_BAD_ARG;
  if (!hip_ok(hipSetDevice(c->device))) return XAAC_FATAL_HIP;
  XaacLimiterParams p = {};
  p.ws_gain = reinterpret_cast<float *>(((uintptr_t)b->workspace + 255) & ~(uintptr_t)255);
  p.ws_flag = reinterpret_cast<int32_t *>(p.ws_gain + (size_t)b->n_streams * 1024);
  p.n_streams = b->n_streams; p.frame_len = b->frame_len; p.num_channels = b->num_channels;
  p.planar = b->planar ? 1 : 0;
  p.samples = b->samples; p.stride = b->stride; p.qshift_adj = b->qshift_adj; p.state = b->state;
  p.pcm16 = b->pcm16; p.status = b->status;
  p.dbg = reinterpret_cast<long long *>(XAAC_DBG_BUF(b->status, b->n_streams)); /* phase timers of -DXL_PROFILE builds */
  if (!hip_ok(xaac_launch_limiter(&p, c->stream))) return XAAC_FATAL_HIP;
  c->last_grid = b->n_streams; c->last_block = 64; c->last_lds = 2 * (XAAC_LIM_MAX_ATTACK + 1024) * 4 + 256;
  return XAAC_OK;
}

int32_t xaac_last_launch(xaac_ctx *c, int32_t *grid, int32_t *block, int32_t *lds_bytes) {
  if (!c) return XAAC_FATAL_NULL_ARG;
  if (grid) *grid = c->last_grid;
  if (block) *block = c->last_block;
  if (lds_bytes) *lds_bytes = c->last_lds;
  return XAAC_OK;
}

}  // extern "C"
