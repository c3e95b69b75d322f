/*
 * limiter_kernel.hip -- ixheaacd_peak_limiter_process (decoder/ixheaacd_peak_limiter.c:201-309) and the
 * round16 hand-off behind it (decoder/ixheaacd_api.c:3676-3681) on gfx950.
 *
 * Mapping: ONE WAVE = ONE STREAM-FRAME.  The limiter is a recursion over the frame's samples (window
 * maximum with the reference's index bookkeeping, then the attack / release smoothing of the gain), so the
 * frame is cut into the three parts that differ in shape:
 *   1. lane-parallel: channel-maximum magnitude of every sample (16 samples per lane, in registers);
 *   2. one uniform instruction stream over the samples: the window maximum exactly as the reference
 *      tracks it (max_idx survives in the state, so its tie-breaking is reproduced: newest on >=, lowest
 *      buffer index on a rescan -- the rescan itself is a wave reduction over the window in LDS) and the
 *      gain recursion of limiter.h; the gain of every sample goes to LDS;
 *   3. lane-parallel, per channel: delayed sample (state delay line for the first attack_time_samples
 *      samples, the frame's own input after that) x gain -> clamp -> WORD32 (in place) / PCM16, and the
 *      frame's last attack_time_samples inputs become the new delay line.
 * All global reads of a channel happen before its writes, so the block is processed in place like the
 * reference does.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "limiter.h"
#include "limiter_kernel.h"

namespace {

#ifdef XL_PROFILE
#define XL_T(i)                                                                        \
  do {                                                                                 \
    if (threadIdx.x == 0) {                                                            \
      long long t_ = clock64();                                                        \
      atomicAdd(reinterpret_cast<unsigned long long *>(p.dbg) + (i), (unsigned long long)(t_ - t_last)); \
      t_last = t_;                                                                     \
    }                                                                                  \
  } while (0)
#else
#define XL_T(i) do { } while (0)
#endif

__device__ __forceinline__ float lane_value(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

/* peak_limiter.c:231-236: lowest index holding the window's maximum (magnitudes: the float order is the
   order of the bit patterns) */
__device__ __forceinline__ void rescan(const float *max_buf, int attack, int lane, int &max_idx, float &cur_max) {
  int best = 0, best_i = 0x7fffffff;
  for (int j = lane; j < attack; j += 64) {
    const int v = __float_as_int(max_buf[j]);
    if (best_i == 0x7fffffff || v > best) {
      best = v;
      best_i = j;
    }
  }
  int top = best_i == 0x7fffffff ? -1 : best;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t = __shfl_xor(top, o);
    top = t > top ? t : top;
  }
  int idx = (best_i != 0x7fffffff && best == top) ? best_i : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t = __shfl_xor(idx, o);
    idx = t < idx ? t : idx;
  }
  max_idx = idx;
  cur_max = __int_as_float(top);
}

}  // namespace

__global__ __launch_bounds__(64) void xaac_limiter_kernel(XaacLimiterParams p) {
  __shared__ float s_max_buf[XAAC_LIM_MAX_ATTACK];
  __shared__ float s_gain[1024];
  const int lane = threadIdx.x, s = blockIdx.x;
  xaac_limiter_state *st = p.state + s;
  int32_t *x = p.samples + (int64_t)s * p.stride;
  const int8_t *qs = p.qshift_adj + (int64_t)s * p.num_channels;
  const int C = p.num_channels, L = p.frame_len;
  const int A = (int)st->attack_time_samples;
  const bool fits = A >= 1 && A <= XAAC_LIM_MAX_ATTACK && (int)st->num_channels == C;
  if (p.status && lane == 0) p.status[s] = fits ? 0 : -1;
  if (!fits) return;
#ifdef XL_PROFILE
  long long t_last = clock64();
#endif

  XlGain g = {st->gain_modified, st->pre_smoothed_gain};
  const float ac = st->attack_constant, rc = st->release_constant;
  const int dii0 = (int)st->delayed_input_index;
  const bool active = xl_active(st->limiter_on, g.pre_smoothed_gain);
  float min_gain = 1.0f;

  if (active) {
    /* ---- 1. channel-maximum magnitudes ---- */
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int i = lane + 64 * k;
      float tmp = 0.0f;
      if (i < L)
        for (int j = 0; j < C; j++) tmp = xl_peak(tmp, x[i * C + j], qs[j]);
      t[k] = tmp;
    }
    for (int i = lane; i < A; i += 64) s_max_buf[i] = st->max_buf[i];
    int max_idx = st->max_idx, cir = st->cir_buf_pnt;
    __syncthreads();
    float cur_max = s_max_buf[max_idx];
    XL_T(0);

    /* ---- 2. window maximum + gain recursion, sample by sample ---- */
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int cnt = L - 64 * k < 64 ? L - 64 * k : 64;
      for (int l = 0; l < cnt; l++) {
        const float tmp = lane_value(t[k], l);
        s_max_buf[cir] = tmp;
        if (max_idx == cir) {
          __syncthreads();
          rescan(s_max_buf, A, lane, max_idx, cur_max);
        } else if (tmp >= cur_max) {
          max_idx = cir;
          cur_max = tmp;
        }
        cir = cir + 1 == A ? 0 : cir + 1;
        const float gain = xl_gain_step(g, xl_target_gain(cur_max), ac, rc);
        s_gain[64 * k + l] = gain;
      }
    }
    __syncthreads();
    XL_T(1);
    for (int i = lane; i < A; i += 64) st->max_buf[i] = s_max_buf[i];
    if (lane == 0) {
      st->max_idx = max_idx;
      st->cir_buf_pnt = cir;
    }
  }

  /* ---- 3. apply to the delayed samples, channel by channel ---- */
  const int end_pos = (dii0 + L) % A; /* delayed_input_index after the frame */
  for (int j = 0; j < C; j++) {
    const int q = qs[j];
    int32_t cur[16];
    float old[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int i = lane + 64 * k;
      cur[k] = 0;
      old[k] = 0.0f;
      if (i < L) {
        cur[k] = x[i * C + j];
        if (i < A) {
          int pos = dii0 + i;
          pos = pos >= A ? pos - A : pos;
          old[k] = st->delayed_input[pos * C + j];
        } else {
          old[k] = xl_scaled(x[(i - A) * C + j], q);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int i = lane + 64 * k;
      if (i < L) {
        int32_t v;
        if (active) {
          const float gain = s_gain[i];
          v = xl_apply(old[k], gain);
          min_gain = gain < min_gain ? gain : min_gain;
        } else {
          v = xl_passthrough(old[k]);
        }
        x[i * C + j] = v;
        if (p.pcm16) p.pcm16[((int64_t)s * L + i) * C + j] = xl_round16(v);
        if (i >= L - A) { /* one of the frame's last attack_time_samples inputs: stays in the delay line */
          int pos = end_pos - (L - i);
          pos = pos < 0 ? pos + A : pos;
          st->delayed_input[pos * C + j] = xl_scaled(cur[k], q);
        }
      }
    }
  }
  XL_T(2);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m = __shfl_xor(min_gain, o);
    min_gain = m < min_gain ? m : min_gain;
  }
  if (lane == 0) {
    st->gain_modified = g.gain_modified;
    st->pre_smoothed_gain = g.pre_smoothed_gain;
    st->delayed_input_index = (uint32_t)end_pos;
    st->min_gain = min_gain;
  }
}

extern "C" hipError_t xaac_launch_limiter(const XaacLimiterParams *p, hipStream_t stream) {
  hipLaunchKernelGGL(xaac_limiter_kernel, dim3(p->n_streams), dim3(64), 0, stream, *p);
  return hipGetLastError();
}
