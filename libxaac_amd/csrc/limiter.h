/*
 * limiter.h -- the arithmetic of ixheaacd_peak_limiter_process (decoder/ixheaacd_peak_limiter.c:201-309),
 * one function per step of its sample loop, for host (oracle, sequential) and gfx950 (limiter_kernel.hip).
 *
 * The reference mixes float and double on purpose-or-not; every conversion below is where the C
 * expression puts it (usual arithmetic conversions, FLT_EVAL_METHOD 0, no contraction -- both compilers
 * get -ffp-contract=off for this code and the pragma below says it again).
 */
#ifndef XAAC_LIMITER_H
#define XAAC_LIMITER_H

#include <stdint.h>

#include "../../include/xaac_amd.h"
#include "fx.h"

#define XL_THR_FIX 2147483647 /* PEAK_LIM_THR_FIX, peak_limiter_struct_def.h:28 */

/* the gain recursion's carried values (peak_limiter.c:216-218) */
struct XlGain {
  float gain_modified;
  double pre_smoothed_gain;
};

/* peak_limiter.c:227-228: tmp = (FLOAT32)MAX(tmp, fabs(sample * gain_t)) */
FX_HD float xl_scaled(int32_t x, int qshift) {
#pragma clang fp contract(off)
  const float gain_t = (float)(int32_t)(1u << qshift);
  return (float)x * gain_t;
}
FX_HD float xl_peak(float tmp, int32_t x, int qshift) {
  float a = xl_scaled(x, qshift);
  a = a < 0.0f ? -a : a; /* fabs; the double round trip is exact */
  return tmp > a ? tmp : a;
}

/* peak_limiter.c:245-249: the target gain for the window maximum */
FX_HD float xl_target_gain(float maximum) {
#pragma clang fp contract(off)
  const float thr = (float)XL_THR_FIX; /* WORD32 -> float in both the compare and the divide: 2^31 */
  return maximum > thr ? thr / maximum : 1.0f;
}

/* peak_limiter.c:251-271: one step of the attack / release smoothing; returns the gain to apply.
   Written as selects (both candidates of each branch are computed): the same values, and on the GPU one
   straight dependency chain instead of four exec-mask regions. */
FX_HD float xl_gain_step(XlGain &g, float gain, float attack_constant, float release_constant) {
#pragma clang fp contract(off)
  const double psg = g.pre_smoothed_gain, gain_d = (double)gain;
  const float cand = (gain - 0.1f * (float)psg) * 1.11111111f;
  const float gm_min = g.gain_modified > cand ? cand : g.gain_modified; /* MIN(x, y) = x > y ? y : x */
  const float gm = gain_d < psg ? gm_min : gain;
  const double gm_d = (double)gm, diff = psg - gm_d;
  double attack = (double)attack_constant * diff + gm_d;
  attack = attack > gain_d ? attack : gain_d; /* MAX */
  const double release = (double)release_constant * diff + gm_d;
  const double next = gm_d < psg ? attack : release;
  g.pre_smoothed_gain = next;
  g.gain_modified = gm;
  return (float)next;
}

/* peak_limiter.c:272-281: delayed sample x gain -> WORD64 (truncation) -> clamp to +-(2^31 - 1) -> WORD32.
   |t| < 2^34, so the WORD64 never overflows; the same result without 64-bit conversions: floats at or above
   2^31 clamp to 2^31 - 1, those at or below -2^31 to -(2^31 - 1) (so does -2^31 + anything truncated), the
   rest convert exactly.  (The int conversion of an out-of-range value is never the selected operand.) */
FX_HD int32_t xl_apply(float delayed, float gain) {
#pragma clang fp contract(off)
  const float t = delayed * gain;
  const float lo = t > -2147483648.0f ? t : -2147483648.0f;
  int32_t v = t >= 2147483648.0f ? 0 : (int32_t)lo;
  v = t >= 2147483648.0f ? XL_THR_FIX : v;
  return v < -XL_THR_FIX ? -XL_THR_FIX : v;
}

/* peak_limiter.c:293 (limiter off and fully released: plain delay): (WORD32)float as x86's cvttss2si does it */
FX_HD int32_t xl_passthrough(float delayed) {
  if (!(delayed < 2147483648.0f && delayed >= -2147483648.0f)) return FX_MIN32;
  return (int32_t)delayed;
}

/* peak_limiter.c:222: which branch the frame takes */
FX_HD int xl_active(uint32_t limiter_on, double pre_smoothed_gain) {
  return limiter_on != 0 || (float)pre_smoothed_gain != 0.0f;
}

/* api.c:3676-3681 */
FX_HD int16_t xl_round16(int32_t v) { return fx_round16(v); }

#endif /* XAAC_LIMITER_H */
