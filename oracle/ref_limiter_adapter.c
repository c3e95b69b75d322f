/*
 * oracle/ref_limiter_adapter.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Drives the reference's own ixheaacd_peak_limiter_init / ixheaacd_peak_limiter_process /
 * ixheaacd_scale_adjust (decoder/ixheaacd_peak_limiter.c) from the boundary struct of include/xaac_amd.h:
 * fills an ia_peak_limiter_struct from a xaac_limiter_state, calls the reference, copies the state back.
 * Linked with the reference objects into oracle/_ref/libref_harness.so (oracle/Makefile.ref).
 */
#include <string.h>
#include "ixheaac_type_def.h"
#include "ixheaac_constants.h"
#include "ixheaac_basic_ops32.h"
#include "ixheaac_basic_ops16.h"
#include "ixheaacd_peak_limiter_struct_def.h"

#include "xaac_amd.h"

WORD32 ixheaacd_peak_limiter_init(ia_peak_limiter_struct *peak_limiter, UWORD32 num_channels, UWORD32 sample_rate,
                                  FLOAT32 *buffer, UWORD32 *delay_in_samples);
VOID ixheaacd_peak_limiter_process(ia_peak_limiter_struct *peak_limiter, VOID *samples, UWORD32 frame_len,
                                   UWORD8 *qshift_adj);

/* boundary state -> the reference's struct (also used by oracle/ref_dropin.c) */
void ref_limiter_to_ref(ia_peak_limiter_struct *r, const xaac_limiter_state *s) {
  const UWORD32 a = s->attack_time_samples, c = s->num_channels;
  memset(r, 0, sizeof(*r));
  r->max_buf = r->buffer;
  r->delayed_input = r->buffer + a * 4 + 32; /* the layout init picks (peak_limiter.c:59) */
  r->attack_time = DEFAULT_ATTACK_TIME_MS;
  r->release_time = DEFAULT_RELEASE_TIME_MS;
  r->attack_constant = s->attack_constant;
  r->release_constant = s->release_constant;
  r->num_channels = c;
  r->attack_time_samples = a;
  r->limiter_on = s->limiter_on;
  r->gain_modified = s->gain_modified;
  r->pre_smoothed_gain = s->pre_smoothed_gain;
  r->delayed_input_index = s->delayed_input_index;
  r->min_gain = s->min_gain;
  r->max_idx = s->max_idx;
  r->cir_buf_pnt = s->cir_buf_pnt;
  memcpy(r->max_buf, s->max_buf, a * sizeof(FLOAT32));
  memcpy(r->delayed_input, s->delayed_input, a * c * sizeof(FLOAT32));
}

void ref_limiter_from_ref(xaac_limiter_state *s, const ia_peak_limiter_struct *r) {
  const UWORD32 a = r->attack_time_samples, c = r->num_channels;
  s->attack_constant = r->attack_constant;
  s->release_constant = r->release_constant;
  s->num_channels = c;
  s->attack_time_samples = a;
  s->limiter_on = r->limiter_on;
  s->gain_modified = r->gain_modified;
  s->pre_smoothed_gain = r->pre_smoothed_gain;
  s->delayed_input_index = r->delayed_input_index;
  s->min_gain = r->min_gain;
  s->max_idx = r->max_idx;
  s->cir_buf_pnt = r->cir_buf_pnt;
  memcpy(s->max_buf, r->max_buf, a * sizeof(FLOAT32));
  memcpy(s->delayed_input, r->delayed_input, a * c * sizeof(FLOAT32));
}

static ia_peak_limiter_struct g_lim; /* 64 KB: not on the stack of a ctypes thread */

int32_t ref_peak_limiter_init(xaac_limiter_state *s, uint32_t num_channels, uint32_t sample_rate) {
  ia_peak_limiter_struct r;
  UWORD32 delay = 0;
  memset(&r, 0, sizeof(r));
  memset(s, 0, sizeof(*s));
  ixheaacd_peak_limiter_init(&r, num_channels, sample_rate, r.buffer, &delay);
  if (delay < 1 || delay > XAAC_LIM_MAX_ATTACK || num_channels > XAAC_LIM_MAX_CH) return -1;
  ref_limiter_from_ref(s, &r);
  return (int32_t)delay;
}

void ref_peak_limiter_process(xaac_limiter_state *s, int32_t *samples, uint32_t frame_len, const int8_t *qshift_adj) {
  ref_limiter_to_ref(&g_lim, s);
  ixheaacd_peak_limiter_process(&g_lim, samples, frame_len, (UWORD8 *)qshift_adj);
  ref_limiter_from_ref(s, &g_lim);
}

/* same shape as xo_peak_limiter_batch / the C ABI: limiter + the round16 loop of api.c:3676-3681 */
void ref_peak_limiter_batch(int32_t n_streams, int32_t frame_len, int32_t num_channels, int32_t *samples,
                            int64_t stride, const int8_t *qshift_adj, xaac_limiter_state *state, int16_t *pcm16) {
  ia_peak_limiter_struct r;
  for (int32_t s = 0; s < n_streams; s++) {
    int32_t *x = samples + (int64_t)s * stride;
    ref_limiter_to_ref(&r, state + s);
    ixheaacd_peak_limiter_process(&r, x, (UWORD32)frame_len, (UWORD8 *)(qshift_adj + (int64_t)s * num_channels));
    ref_limiter_from_ref(state + s, &r);
    if (pcm16)
      for (int32_t i = 0; i < frame_len * num_channels; i++)
        pcm16[(int64_t)s * frame_len * num_channels + i] = ixheaac_round16(x[i]);
  }
}
