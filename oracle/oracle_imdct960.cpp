/*
 * oracle_imdct960.cpp -- TEST INFRASTRUCTURE: the 960-line AAC IMDCT + windowing / overlap-add on the CPU, i.e.
 * libxaac_amd/csrc/imdct960.h (the restatement of the frame_length == 960 branches of ixheaacd_imdct_process,
 * decoder/ixheaacd_lpfuncs.c:347-802, with decoder/ixheaacd_aac_imdct.c:1624-2742) compiled for the host and run
 * sequentially.  Pinned to the compiled reference by tests/test_imdct960_oracle_vs_reference.py (ref_imdct960_process in
 * oracle/ref_harness.c calls the reference's own function).  Only tests/, __graft_entry__.smoke() and bench.py's checker
 * legs use it; the product never does.
 */
#include <stdint.h>
#include <string.h>

#include "../libxaac_amd/csrc/imdct960.h"

extern "C" {

/* spec[960] (not modified), overlap[480] in/out, prev_seq / prev_shape in/out, out[960] at stride s; returns qshift_adj or
   -1 for a window_sequence / window_shape the bitstream fields cannot carry */
int xo_imdct960_process(const int32_t *spec, int32_t *ovl, int16_t *prev_seq, int16_t *prev_shape, int seq, int shape,
                        int32_t *out, int s) {
  if ((unsigned)seq > 3 || (unsigned)shape > 1 || (unsigned)*prev_seq > 3 || (unsigned)*prev_shape > 1) return -1;
  int32_t y[960], a[960], old[480];
  memcpy(old, ovl, sizeof(old));
  int32_t acc = 0;
  for (int i = 0; i < 960; i++) acc |= fx_abs_nrm(spec[i]);
  const bool edge = *prev_seq == X9_LONG_START || *prev_seq == X9_EIGHT_SHORT;
  X9Sink sk = {out, nullptr, s, x9_qshift_adj(seq, edge), 0};
  x9_imdct_process(spec, old, ovl, y, a, fx_norm32(acc), seq, shape, *prev_seq, *prev_shape, sk, 0, 1);
  *prev_seq = (int16_t)seq;
  *prev_shape = (int16_t)shape;
  return sk.qadj;
}

/* nch channel-frames: spec[nch][960], ovl[nch][480], state[nch][2] = (window_sequence, window_shape) bytes, ics[nch][2],
   out32[nch][960] and / or pcm[nch][960] */
void xo_imdct960_batch(int nch, const int32_t *spec, int32_t *ovl, uint8_t *state, const uint8_t *ics, int32_t *out32,
                       int16_t *pcm, int8_t *qadj, int pcm_mode) {
  int32_t tmp[960];
  for (int c = 0; c < nch; c++) {
    int16_t ps = state[2 * c], pw = state[2 * c + 1];
    int32_t *o = out32 ? out32 + 960 * (size_t)c : tmp;
    const int qa = xo_imdct960_process(spec + 960 * (size_t)c, ovl + 480 * (size_t)c, &ps, &pw, ics[2 * c], ics[2 * c + 1], o, 1);
    if (qa < 0) continue;
    state[2 * c] = (uint8_t)ps;
    state[2 * c + 1] = (uint8_t)pw;
    if (qadj) qadj[c] = (int8_t)qa;
    if (pcm)
      for (int i = 0; i < 960; i++)
        pcm[960 * (size_t)c + i] = fx_round16(pcm_mode ? fx_shl_sat(o[i], qa) : fx_shlw(o[i], qa));
  }
}

}  // extern "C"

/* stage probes for tests: the pre twiddle alone (natural order), the 480-point transform alone */
extern "C" void xo_i960_pre_twiddle(const int32_t *spec, int sh, int32_t *z) {
  for (int c = 0; c < 480; c++) x9_st(z, c, x9_pre_twiddle<true>(spec, c, sh));
}
