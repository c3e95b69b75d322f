/*
 * limiter_kernel.hip -- ixheaacd_peak_limiter_process (decoder/ixheaacd_peak_limiter.c:201-309) and the
 * round16 hand-off behind it (decoder/ixheaacd_api.c:3676-3681) on gfx950.
 *
 * Mapping: ONE WAVE = ONE STREAM-FRAME.  The reference walks the frame sample by sample: window maximum
 * (a circular buffer plus the index of its largest entry), target gain, attack / release smoothing, apply
 * to the delayed sample.  Here the walk is cut by what really is a recursion:
 *   1. lane-parallel: channel-maximum magnitude of every sample; with the attack_time_samples magnitudes of
 *      the state in front they form the time-ordered array W (LDS);
 *   2. lane-parallel: the window maximum of every sample = sliding maximum over W (log2 doubling rounds in
 *      LDS, exact: max is idempotent), from it the target gain (the divide) and two bit masks per 64 samples:
 *      "this sample becomes the tracked maximum" (its magnitude equals its window maximum) and "this sample's
 *      target gain is below 1";
 *   3. the reference's max_idx bookkeeping survives in the state, so it is reproduced -- as events, not per
 *      sample: inside a 64-sample chunk the tracked element is the last flagged sample unless the tracked one
 *      leaves the window first, which is the reference's rescan (lowest buffer index among the window's
 *      maxima: a wave reduction over W);
 *   4. the gain smoothing is the one true per-sample recursion (float/double mix of limiter.h); it runs as a
 *      uniform instruction stream, but only from the first sample that asks for limiting: a released limiter
 *      (pre_smoothed_gain == 1.0 exactly) with target gain 1 is a fixed point of the recursion;
 *   5. lane-parallel, per channel: delayed sample (state delay line for the first attack_time_samples samples,
 *      the frame's own input after that) x gain -> clamp -> WORD32 (in place) / PCM16; the frame's last
 *      attack_time_samples inputs become the new delay line, W's tail the new window.
 * A state whose max_idx does not point at its window's maximum (it cannot come from init + process, but the
 * reference would still run on it), and windows shorter than a chunk (rates below 12.8 kHz), take the plain
 * per-sample walk (`walk`), which is the reference's loop verbatim on W.
 * All global reads of a channel happen before its writes, so the block is processed in place like the
 * reference does.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "limiter.h"
#include "limiter_kernel.h"

namespace {

#ifdef XL_PROFILE
#define XL_T(i)                                                                                          \
  do {                                                                                                   \
    if (threadIdx.x == 0) {                                                                              \
      long long t_ = clock64();                                                                          \
      atomicAdd(reinterpret_cast<unsigned long long *>(p.dbg) + (i), (unsigned long long)(t_ - t_last)); \
      t_last = t_;                                                                                       \
    }                                                                                                    \
  } while (0)
#else
#define XL_T(i) do { } while (0)
#endif

constexpr int kMaxW = XAAC_LIM_MAX_ATTACK + 1024;

__device__ __forceinline__ float lane_value(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

/* peak_limiter.c:231-236 at step `step`: the window is W[step + 1 .. step + A]; the reference scans buffer
   positions upwards and keeps the first largest one.  Magnitudes are >= 0: float order = bit-pattern order.
   Returns the W index of the new tracked element. */
__device__ __forceinline__ int rescan(const float *w, int step, int A, int cir0, int lane) {
  int best = -1, best_pos = 0x7fffffff;
  int pos = (cir0 + step + 1 + lane) % A; /* buffer position of W[step + 1 + lane] */
  const int adv = 64 % A;
  for (int j = lane; j < A; j += 64) {
    const int v = __float_as_int(w[step + 1 + j]);
    if (v > best || (v == best && pos < best_pos)) {
      best = v;
      best_pos = pos;
    }
    pos += adv;
    pos = pos >= A ? pos - A : pos;
  }
  int top = best;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t = __shfl_xor(top, o);
    top = t > top ? t : top;
  }
  int p = best == top ? best_pos : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t = __shfl_xor(p, o);
    p = t < p ? t : p;
  }
  p = uni(p);
  /* back to the W index: the element of the window that sits at buffer position p */
  int first = (cir0 + step + 1) % A;
  int d = p - first;
  d = d < 0 ? d + A : d;
  return step + 1 + d;
}

/* N consecutive channels of one sample as one access */
template <int N, typename T>
__device__ __forceinline__ void ld(const T *p, T (&v)[N]) {
  if (N == 2 && sizeof(T) == 4) {
    const int2 t = *reinterpret_cast<const int2 *>(p);
    __builtin_memcpy(&v[0], &t.x, 4);
    __builtin_memcpy(&v[1], &t.y, 4);
  } else {
#pragma unroll
    for (int c = 0; c < N; c++) v[c] = p[c];
  }
}
template <int N, typename T>
__device__ __forceinline__ void st_(T *p, const T (&v)[N]) {
  if (N == 2 && sizeof(T) == 4) {
    int2 t;
    __builtin_memcpy(&t.x, &v[0], 4);
    __builtin_memcpy(&t.y, &v[1], 4);
    *reinterpret_cast<int2 *>(p) = t;
  } else if (N == 2 && sizeof(T) == 2) {
    uint32_t t;
    __builtin_memcpy(&t, &v[0], 4);
    *reinterpret_cast<uint32_t *>(p) = t;
  } else {
#pragma unroll
    for (int c = 0; c < N; c++) p[c] = v[c];
  }
}

}  // namespace

/* CT: channel count known at compile time (1, 2: the loads of a phase are then all in flight together), 0: any */
template <int CT>
__global__ __launch_bounds__(64) void xaac_limiter_kernel(XaacLimiterParams p) {
  __shared__ float s_w[kMaxW]; /* W: the state's window in time order, then the frame's magnitudes */
  __shared__ float s_g[kMaxW]; /* sliding-maximum workspace; s_g[A + i] ends as the gain of sample i */
  __shared__ float s_run[64];
  const int lane = threadIdx.x, s = blockIdx.x;
  xaac_limiter_state *st = p.state + s;
  int32_t *x = p.samples + (int64_t)s * p.stride;
  const int8_t *qs = p.qshift_adj + (int64_t)s * p.num_channels;
  const int C = CT ? CT : p.num_channels, L = p.frame_len;
  const int A = (int)st->attack_time_samples;
  const bool fits = A >= 1 && A <= XAAC_LIM_MAX_ATTACK && (int)st->num_channels == C;
#ifndef XL_PROFILE
  if (p.status && lane == 0) p.status[s] = fits ? 0 : -1;
#endif
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
    const int cir0 = st->cir_buf_pnt, n = A + L;
    /* ---- 1. W ---- */
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int i = lane + 64 * k;
      float tmp = 0.0f;
      if (i < L) {
        if (CT == 2) {
          const int2 v = reinterpret_cast<const int2 *>(x)[i];
          tmp = xl_peak(xl_peak(tmp, v.x, qs[0]), v.y, qs[1]);
        } else {
          for (int j = 0; j < C; j++) tmp = xl_peak(tmp, x[i * C + j], qs[j]);
        }
        s_w[A + i] = tmp;
      }
      t[k] = tmp;
    }
    for (int k = lane; k < A; k += 64) {
      int pos = cir0 + k;
      pos = pos >= A ? pos - A : pos;
      s_w[k] = st->max_buf[pos];
    }
    int d0 = cir0 - st->max_idx;
    d0 = d0 < 0 ? d0 + A : d0;
    int cur = d0 == 0 ? 0 : A - d0; /* W index of the tracked maximum (an element leaves at step = its index) */
    __syncthreads();
    XL_T(0);

    /* ---- 2. window maxima: mx[i] = max W[i + 1 .. i + A] ----
       Lane l owns the run W[25 l .. 25 l + 24] (25: odd, so run-strided LDS accesses do not collide).  A window
       (A >= 64 > two runs) is the tail of the run it starts in, some whole runs, and the head of the run it
       ends in: suffix maxima of every run go to LDS, the run maxima too, prefix maxima stay in registers. */
    float mx[16], hist_max = 0.0f;
    if (A >= 64) {
      constexpr int RL = 25;
      float pre[RL];
#pragma unroll
      for (int r = 0; r < RL; r++) {
        const int xi = RL * lane + r;
        pre[r] = xi < n ? s_w[xi] : 0.0f;
      }
      float run = 0.0f;
#pragma unroll
      for (int r = RL - 1; r >= 0; r--) {
        const int xi = RL * lane + r;
        run = pre[r] > run ? pre[r] : run;
        if (xi < n) s_g[xi] = run; /* max W[xi .. end of the run] */
      }
#pragma unroll
      for (int r = 1; r < RL; r++) pre[r] = pre[r] > pre[r - 1] ? pre[r] : pre[r - 1];
      s_run[lane] = run;
      __syncthreads();
      /* whole runs between the window's first run and this one: the window of output r starts at y0 + r */
      const int y0 = RL * lane - A + 1;
      const int ra = (y0 + RL * 64) / RL - 64; /* floor(y0 / RL) */
      float q2 = 0.0f;                         /* max of runs ra + 2 .. lane - 1 */
      for (int d = 1; d <= A / RL + 1; d++) {
        const int j = lane - d;
        if (j >= ra + 2 && j >= 0) {
          const float v = s_run[j];
          q2 = v > q2 ? v : q2;
        }
      }
      float q1 = q2;                           /* ... of runs ra + 1 .. lane - 1 */
      if (ra + 1 >= 0 && ra + 1 < lane) {
        const float v = s_run[ra + 1];
        q1 = v > q1 ? v : q1;
      }
#pragma unroll
      for (int r = 0; r < RL; r++) {
        const int xi = RL * lane + r, y = y0 + r;
        float o = 0.0f;
        if (y >= 0 && xi < n) {
          const float tail = s_g[y];
          const float mid = y >= RL * (ra + 1) ? q2 : q1; /* the window starts in run ra + 1 : in run ra */
          o = tail > mid ? tail : mid;
          o = pre[r] > o ? pre[r] : o;
        }
        pre[r] = o;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < RL; r++) {
        const int xi = RL * lane + r;
        if (xi >= A - 1 && xi < n) s_g[xi] = pre[r]; /* window maximum of the window that ENDS at W[xi] */
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int i = lane + 64 * k;
        mx[k] = i < L ? s_g[A + i] : 0.0f;
      }
      hist_max = s_g[A - 1];
    } else {
#pragma unroll
      for (int k = 0; k < 16; k++) mx[k] = 0.0f;
    }
    const bool fast = A >= 64 && s_w[cur] == hist_max;
    __syncthreads();
    XL_T(1);

    if (fast) {
      unsigned long long fm[16], lm[16];
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int i = lane + 64 * k;
        float tg = 1.0f;
        if (__ballot(mx[k] > (float)XL_THR_FIX)) tg = xl_target_gain(mx[k]);
        fm[k] = __ballot(i < L && t[k] == mx[k]);
        lm[k] = __ballot(i < L && tg < 1.0f);
        if (i < L) s_g[A + i] = tg;
      }
      __syncthreads();
      /* ---- 3. which element is tracked when the frame ends ---- */
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int base = 64 * k, lim = L - base < 64 ? L - base : 64;
        if (lim <= 0) break;
        int done = 0; /* steps of the chunk already walked */
        for (;;) {
          const unsigned long long rest = fm[k] >> done << done;
          const int rel = cur - base; /* the tracked element leaves at this step of the chunk */
          if (rel >= lim || (rest & ((1ull << rel) - 1ull)) != 0) {
            /* it outlives the chunk, or a flagged sample takes over first: from there on every flagged sample
               takes over in turn (A >= 64: none of them can leave inside the chunk) */
            if (rest) cur = A + base + 63 - __clzll((long long)rest);
            break;
          }
          cur = rescan(s_w, base + rel, A, cir0, lane);
          done = rel + 1;
          if (done >= lim) break;
        }
      }
      XL_T(2);
      /* ---- 4. gain smoothing, from the first sample that wants limiting ---- */
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int base = 64 * k, lim = L - base < 64 ? L - base : 64;
        if (lim <= 0) break;
        int l = 0;
        if (g.pre_smoothed_gain == 1.0) { /* fixed point while the target stays 1 (a step there leaves gain_modified = 1) */
          l = lm[k] ? (int)__builtin_ctzll(lm[k]) : lim;
          if (l > 0) g.gain_modified = 1.0f;
          if (l >= lim) continue;
        }
        const float tg = s_g[A + base + lane];
        for (; l < lim; l++) s_g[A + base + l] = xl_gain_step(g, lane_value(tg, l), ac, rc);
      }
      __syncthreads();
      XL_T(3);
    } else {
      /* the reference's per-sample walk */
      float cur_max = s_w[cur];
      for (int i = 0; i < L; i++) {
        const float tmp = s_w[A + i];
        if (cur == i) {
          cur = rescan(s_w, i, A, cir0, lane);
          cur_max = s_w[cur];
        } else if (tmp >= cur_max) {
          cur = A + i;
          cur_max = tmp;
        }
        s_g[A + i] = xl_gain_step(g, xl_target_gain(cur_max), ac, rc);
      }
      __syncthreads();
      XL_T(4);
    }

    /* the window the next frame starts from */
    int end_cir = (cir0 + L) % A;
    for (int k = lane; k < A; k += 64) {
      int pos = end_cir + k; /* W[L + k] is the k-th oldest of the new window */
      pos = pos >= A ? pos - A : pos;
      st->max_buf[pos] = s_w[L + k];
    }
    if (lane == 0) {
      int pos = end_cir + (cur - L);
      st->max_idx = pos >= A ? pos - A : pos;
      st->cir_buf_pnt = end_cir;
    }
  }

  /* ---- 5. apply to the delayed samples, N channels a pass ---- */
  constexpr int N = CT == 2 ? 2 : 1;
  const int end_pos = (dii0 + L) % A; /* delayed_input_index after the frame */
  for (int j = 0; j < C; j += N) {
    int q[N];
#pragma unroll
    for (int c = 0; c < N; c++) q[c] = qs[j + c];
    /* the delay line's samples first: the frame's tail overwrites their slots */
    float line[8][N];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = lane + 64 * k;
      int pos = dii0 + i;
      pos = pos >= A ? pos - A : pos;
#pragma unroll
      for (int c = 0; c < N; c++) line[k][c] = 0.0f;
      if (i < A && i < L) ld<N>(st->delayed_input + pos * C + j, line[k]);
    }
#pragma unroll
    for (int half = 1; half >= 0; half--) { /* upper half first: it still needs the lower half's input */
      int32_t now[8][N], before[8][N];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int i = lane + 64 * (k + 8 * half);
#pragma unroll
        for (int c = 0; c < N; c++) now[k][c] = before[k][c] = 0;
        if (i < L) {
          ld<N>(x + i * C + j, now[k]);
          if (i >= A) ld<N>(x + (i - A) * C + j, before[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int i = lane + 64 * (k + 8 * half);
        if (i < L) {
          const float gain = active ? s_g[A + i] : 1.0f;
          int32_t v[N];
          int16_t v16[N];
          float keep[N];
#pragma unroll
          for (int c = 0; c < N; c++) {
            float old = xl_scaled(before[k][c], q[c]);
            if (half == 0 && i < A) old = line[k][c];
            v[c] = active ? xl_apply(old, gain) : xl_passthrough(old);
            v16[c] = xl_round16(v[c]);
            keep[c] = xl_scaled(now[k][c], q[c]);
          }
          if (active) min_gain = gain < min_gain ? gain : min_gain;
          st_<N>(x + i * C + j, v);
          if (p.pcm16) st_<N>(p.pcm16 + ((int64_t)s * L + i) * C + j, v16);
          if (i >= L - A) { /* one of the frame's last attack_time_samples inputs: stays in the delay line */
            int pos = end_pos - (L - i);
            pos = pos < 0 ? pos + A : pos;
            st_<N>(st->delayed_input + pos * C + j, keep);
          }
        }
      }
    }
  }
  XL_T(5);
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
  if (p->num_channels == 1)
    hipLaunchKernelGGL(xaac_limiter_kernel<1>, dim3(p->n_streams), dim3(64), 0, stream, *p);
  else if (p->num_channels == 2)
    hipLaunchKernelGGL(xaac_limiter_kernel<2>, dim3(p->n_streams), dim3(64), 0, stream, *p);
  else
    hipLaunchKernelGGL(xaac_limiter_kernel<0>, dim3(p->n_streams), dim3(64), 0, stream, *p);
  return hipGetLastError();
}
