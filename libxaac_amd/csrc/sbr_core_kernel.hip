/*
 * sbr_core_kernel.hip -- the serial middle of ixheaacd_sbr_dec (decoder/ixheaacd_sbr_dec.c:726-775,
 * :1050-1245, :1283-1308; low-power mode) on gfx950: overlap restore + xs_rescale_x_overlap, block
 * floating point, LPP transposer, envelope adjustment, LPC/overlap/scale state update -- everything
 * between the two QMF banks (which run as their own slot-parallel kernels before and after).
 *
 * First mapping (round 1): ONE LANE = ONE CHANNEL running the scalar code of sbr_core.h on the
 * channel's 40 x 64 QMF matrix in the workspace.  The control flow of this stage is data dependent per
 * channel (envelope count, limiter bands, alias groups), channels are plentiful (16384 per batch), and
 * sharing the scalar source with the oracle keeps it bit-exact by construction.  Its cost -- lanes
 * walk 10 KB-strided matrices, i.e. uncoalesced -- is the known next optimisation (DESIGN.md §7:
 * band-parallel energy / covariance / gain application with the matrix staged in LDS).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sbr_core.h"
#include "sbr_core_kernel.h"

__global__ __launch_bounds__(64) void xaac_sbr_core_lp_kernel(XaacSbrCoreParams p) {
  const int ch = blockIdx.x * 64 + threadIdx.x;
  if (ch >= p.n_ch) return;
  const xaac_sbr_header *h = p.header + ch;
  const xaac_sbr_frame *f = p.frame + ch;
  xaac_sbr_state *st = p.state + ch;
  XsQmf x = {p.x + (size_t)ch * XAAC_SBR_X_WORDS, 1};
  for (int l = 0; l < 6; l++)
    for (int k = 0; k < 64; k++) x(l, k) = st->overlap[64 * l + k];
  st->lb_scale = 0;
  if (f->apply_processing) xs_rescale_x_overlap(h, f, st, x);
  /* what ixheaacd_cplx_anal_qmffilt leaves in the scale struct (generic:630-631) */
  st->st_lb_scale = 0;
  st->lb_scale = -10;
  int save_lb_scale = 0;
  const int rc = xs_sbr_core_lp(h, f, st, x, &save_lb_scale);
  int16_t *par = p.syn_par + 8 * (size_t)ch;
  par[0] = st->lb_scale;
  par[1] = st->ov_lb_scale;
  par[2] = st->hb_scale;
  par[3] = st->st_syn_scale;
  par[4] = st->syn_lsb;
  par[5] = st->syn_usb;
  for (int l = 0; l < 6; l++)
    for (int k = 0; k < 64; k++) st->overlap[64 * l + k] = x(32 + l, k);
  st->ov_lb_scale = (int16_t)save_lb_scale;
  if (p.status) p.status[ch] = rc;
}

extern "C" hipError_t xaac_launch_sbr_core_lp(const XaacSbrCoreParams *p, hipStream_t stream) {
  const int grid = (p->n_ch + 63) / 64;
  hipLaunchKernelGGL(xaac_sbr_core_lp_kernel, dim3(grid), dim3(64), 0, stream, *p);
  return hipGetLastError();
}
