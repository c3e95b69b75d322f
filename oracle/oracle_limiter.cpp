/*
 * oracle_limiter.cpp -- CPU restatement of the AAC-LC post stage: ixheaacd_peak_limiter_init /
 * ixheaacd_peak_limiter_process (decoder/ixheaacd_peak_limiter.c:46-77, :201-309) and the round16 loop
 * behind it (decoder/ixheaacd_api.c:3676-3681).
 *
 * TEST INFRASTRUCTURE (see oracle/Makefile): the sequential sample loop of the reference, written over
 * the step functions of libxaac_amd/csrc/limiter.h that the GPU kernel also uses.  Pinned against the
 * reference's own function through oracle/_ref/libref_harness.so (ref_peak_limiter_process,
 * tests/test_limiter_oracle_vs_reference.py) and the golden records of tests/golden/.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../libxaac_amd/csrc/limiter.h"

extern "C" {

/* peak_limiter.c:46-77; returns the delay (attack_time_samples), 0 if the limiter would stay uninitialised */
int32_t xo_peak_limiter_init(xaac_limiter_state *s, uint32_t num_channels, uint32_t sample_rate) {
  const uint32_t attack = (uint32_t)(5.0f * sample_rate / 1000);
  memset(s, 0, sizeof(*s));
  if (attack < 1) return 0;
  s->attack_time_samples = attack;
  s->attack_constant = (float)pow(0.1, 1.0 / (attack + 1));
  s->release_constant = (float)pow(0.1, 1.0 / (50.0f * sample_rate / 1000 + 1));
  s->num_channels = num_channels;
  s->min_gain = 1.0f;
  s->limiter_on = 1;
  s->pre_smoothed_gain = 1.0f;
  s->gain_modified = 1.0f;
  return (int32_t)attack;
}

/* peak_limiter.c:201-309 on one stream's interleaved WORD32 block */
void xo_peak_limiter_process(xaac_limiter_state *s, int32_t *samples, uint32_t frame_len, const int8_t *qshift_adj) {
  const uint32_t nch = s->num_channels, attack = s->attack_time_samples;
  XlGain g = {s->gain_modified, s->pre_smoothed_gain};
  uint32_t dii = s->delayed_input_index;
  float min_gain = 1.0f;

  if (xl_active(s->limiter_on, s->pre_smoothed_gain)) {
    for (uint32_t i = 0; i < frame_len; i++) {
      float tmp = 0.0f;
      for (uint32_t j = 0; j < nch; j++) tmp = xl_peak(tmp, samples[i * nch + j], qshift_adj[j]);
      s->max_buf[s->cir_buf_pnt] = tmp;
      if (s->max_idx == s->cir_buf_pnt) { /* the maximum just left the window: rescan (:231-236) */
        s->max_idx = 0;
        for (uint32_t j = 1; j < attack; j++)
          if (s->max_buf[j] > s->max_buf[s->max_idx]) s->max_idx = (int32_t)j;
      } else if (tmp >= s->max_buf[s->max_idx]) {
        s->max_idx = s->cir_buf_pnt;
      }
      if (++s->cir_buf_pnt == (int32_t)attack) s->cir_buf_pnt = 0;

      const float gain = xl_gain_step(g, xl_target_gain(s->max_buf[s->max_idx]), s->attack_constant, s->release_constant);
      for (uint32_t j = 0; j < nch; j++) {
        const float delayed = s->delayed_input[dii * nch + j];
        s->delayed_input[dii * nch + j] = xl_scaled(samples[i * nch + j], qshift_adj[j]);
        samples[i * nch + j] = xl_apply(delayed, gain);
      }
      if (++dii >= attack) dii = 0;
      if (gain < min_gain) min_gain = gain;
    }
  } else {
    for (uint32_t i = 0; i < frame_len; i++) {
      for (uint32_t j = 0; j < nch; j++) {
        const float delayed = s->delayed_input[dii * nch + j];
        s->delayed_input[dii * nch + j] = xl_scaled(samples[i * nch + j], qshift_adj[j]);
        samples[i * nch + j] = xl_passthrough(delayed);
      }
      if (++dii >= attack) dii = 0;
    }
  }
  s->gain_modified = g.gain_modified;
  s->pre_smoothed_gain = g.pre_smoothed_gain;
  s->delayed_input_index = dii;
  s->min_gain = min_gain;
}

/* the batch the C ABI takes (bench.py's cpu_baseline loop and the tests): limiter + round16 */
void xo_peak_limiter_batch(int32_t n_streams, int32_t frame_len, int32_t num_channels, int32_t *samples, int64_t stride,
                           const int8_t *qshift_adj, xaac_limiter_state *state, int16_t *pcm16) {
  for (int32_t s = 0; s < n_streams; s++) {
    int32_t *x = samples + (int64_t)s * stride;
    xo_peak_limiter_process(state + s, x, (uint32_t)frame_len, qshift_adj + (int64_t)s * num_channels);
    if (pcm16)
      for (int32_t i = 0; i < frame_len * num_channels; i++)
        pcm16[(int64_t)s * frame_len * num_channels + i] = xl_round16(x[i]);
  }
}

}  /* extern "C" */
