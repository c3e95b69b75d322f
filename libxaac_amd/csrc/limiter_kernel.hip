/*
 * limiter_kernel.hip -- ixheaacd_peak_limiter_process (decoder/ixheaacd_peak_limiter.c:201-309) and the
 * round16 hand-off behind it (decoder/ixheaacd_api.c:3676-3681) on gfx950.
 *
 * The reference walks a frame sample by sample: window maximum (a circular buffer plus the index of its largest
 * entry), target gain, attack / release smoothing, apply to the delayed sample.  Only the smoothing really is a
 * recursion, and it is one per STREAM -- so the work is cut into three launches with two mappings:
 *
 * xaac_limiter_front_kernel, one WAVE per stream-frame:
 *   1. lane-parallel: channel-maximum magnitude of every sample; with the attack_time_samples magnitudes of the
 *      state in front they form the time-ordered array W (LDS);
 *   2. lane-parallel: the window maximum of every sample (sliding maximum over W by run prefix / suffix maxima,
 *      exact: max is idempotent), from it the target gain (the divide, only where something is over the
 *      threshold) and two bit masks per 64 samples: "this sample becomes the tracked maximum" (its magnitude
 *      equals its window maximum) and "this sample's target gain is below 1";
 *   3. the reference's max_idx bookkeeping survives in the state, so it is reproduced -- as events, not per
 *      sample: inside a 64-sample chunk the tracked element is the last flagged sample unless the tracked one
 *      leaves the window first, which is the reference's rescan (lowest buffer index among the window's maxima:
 *      a wave reduction over W);
 *   4. a released limiter (pre_smoothed_gain == 1.0 exactly) whose frame never asks for limiting is a fixed point
 *      of the smoothing: all gains are 1, the frame is finished here (step 6).  Otherwise the target gains go to
 *      the workspace, with the first sample the recursion has to start at.
 * xaac_limiter_gain_kernel, one LANE per stream:
 *   5. the smoothing recursion of limiter.h (float / double mix) over the target gains of 64 streams at once --
 *      one instruction stream per 64 streams instead of per stream; gains replace the target gains in place.
 * xaac_limiter_apply_kernel, one WAVE per stream-frame that still needs it:
 *   6. lane-parallel, N channels a pass: delayed sample (state delay line for the first attack_time_samples
 *      samples, the frame's own input after that) x gain -> clamp -> WORD32 (in place) / PCM16; the frame's last
 *      attack_time_samples inputs become the new delay line.  All global reads of a pass happen before its
 *      writes, so the block is processed in place like the reference does.
 * A state whose max_idx does not point at its window's maximum (it cannot come from init + process, but the
 * reference would still run on it), and windows shorter than a chunk (rates below 12.8 kHz), take the plain
 * per-sample walk in step 2-3, which is the reference's loop verbatim on W.
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


/* N channels of sample i of a stream's WORD32 block: interleaved [i][channel] (one access) or planar [channel][i] */
template <int N>
__device__ __forceinline__ void ld_block(const int32_t *x, int i, int j, int C, int L, bool planar, int32_t (&v)[N]) {
  if (planar) {
#pragma unroll
    for (int c = 0; c < N; c++) v[c] = x[(j + c) * L + i];
  } else {
    ld<N>(x + i * C + j, v);
  }
}
template <int N>
__device__ __forceinline__ void st_block(int32_t *x, int i, int j, int C, int L, bool planar, const int32_t (&v)[N]) {
  if (planar) {
#pragma unroll
    for (int c = 0; c < N; c++) x[(j + c) * L + i] = v[c];
  } else {
    st_<N>(x + i * C + j, v);
  }
}

/* ---- step 6: gains from `gain_of(i)` applied to the delayed samples of stream s ---- */
template <int CT, typename GainOf>
__device__ __forceinline__ void apply_frame(const XaacLimiterParams &p, int s, int lane, bool active, GainOf gain_of) {
  xaac_limiter_state *st = p.state + s;
  int32_t *x = p.samples + (int64_t)s * p.stride;
  const int8_t *qs = p.qshift_adj + (int64_t)s * p.num_channels;
  const int C = CT ? CT : p.num_channels, L = p.frame_len;
  const int A = (int)st->attack_time_samples, dii0 = (int)st->delayed_input_index;
  const bool planar = p.planar != 0;
  constexpr int N = CT == 2 ? 2 : 1;
  const int end_pos = (dii0 + L) % A; /* delayed_input_index after the frame */
  float min_gain = 1.0f;
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
      float gain[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int i = lane + 64 * (k + 8 * half);
#pragma unroll
        for (int c = 0; c < N; c++) now[k][c] = before[k][c] = 0;
        gain[k] = 1.0f;
        if (i < L) {
          ld_block<N>(x, i, j, C, L, planar, now[k]);
          if (i >= A) ld_block<N>(x, i - A, j, C, L, planar, before[k]);
          gain[k] = gain_of(i);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int i = lane + 64 * (k + 8 * half);
        if (i < L) {
          int32_t v[N];
          int16_t v16[N];
          float keep[N];
#pragma unroll
          for (int c = 0; c < N; c++) {
            float old = xl_scaled(before[k][c], q[c]);
            if (half == 0 && i < A) old = line[k][c];
            v[c] = active ? xl_apply(old, gain[k]) : xl_passthrough(old);
            v16[c] = xl_round16(v[c]);
            keep[c] = xl_scaled(now[k][c], q[c]);
          }
          if (active) min_gain = gain[k] < min_gain ? gain[k] : min_gain;
          st_block<N>(x, i, j, C, L, planar, v);
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
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m = __shfl_xor(min_gain, o);
    min_gain = m < min_gain ? m : min_gain;
  }
  if (lane == 0) {
    st->delayed_input_index = (uint32_t)end_pos;
    st->min_gain = min_gain;
  }
}

}  // namespace

/* CT: channel count known at compile time (1, 2: the loads of a phase are then all in flight together), 0: any */
template <int CT>
__global__ __launch_bounds__(64) void xaac_limiter_front_kernel(XaacLimiterParams p) {
  __shared__ float s_w[kMaxW]; /* W: the state's window in time order, then the frame's magnitudes */
  __shared__ float s_g[kMaxW]; /* run suffix maxima, then window maxima; the walk's target gains */
  __shared__ float s_run[64];
  const int lane = threadIdx.x, s = blockIdx.x;
  xaac_limiter_state *st = p.state + s;
  const int32_t *x = p.samples + (int64_t)s * p.stride;
  const int8_t *qs = p.qshift_adj + (int64_t)s * p.num_channels;
  const int C = CT ? CT : p.num_channels, L = p.frame_len;
  const int A = (int)st->attack_time_samples;
  const bool fits = A >= 1 && A <= XAAC_LIM_MAX_ATTACK && (int)st->num_channels == C;
#ifndef XL_PROFILE
  if (p.status && lane == 0) p.status[s] = fits ? 0 : -1;
#endif
  if (!fits) {
    if (lane == 0) p.ws_flag[2 * s] = 1; /* nothing to do for the other two kernels */
    return;
  }
#ifdef XL_PROFILE
  long long t_last = clock64();
#endif
  const double psg0 = st->pre_smoothed_gain;
  const bool active = xl_active(st->limiter_on, psg0);
  if (!active) { /* peak_limiter.c:288-300: a plain delay */
    if (lane == 0) p.ws_flag[2 * s] = 1;
    apply_frame<CT>(p, s, lane, false, [](int) { return 1.0f; });
    return;
  }

  const int cir0 = st->cir_buf_pnt, n = A + L;
  /* ---- 1. W ---- */
  float t[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int i = lane + 64 * k;
    float tmp = 0.0f;
    if (i < L) {
      if (p.planar) {
        for (int j = 0; j < C; j++) tmp = xl_peak(tmp, x[j * L + i], qs[j]);
      } else if (CT == 2) {
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
      if (xi >= A - 1 && xi < n) s_g[xi] = pre[r]; /* maximum of the window that ENDS at W[xi] */
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

  float tg[16]; /* target gain of sample lane + 64 k */
  int first_lim = L; /* first sample whose target gain is below 1 */
  if (fast) {
    unsigned long long fm[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int i = lane + 64 * k;
      tg[k] = 1.0f;
      if (__ballot(mx[k] > (float)XL_THR_FIX)) tg[k] = xl_target_gain(mx[k]);
      fm[k] = __ballot(i < L && t[k] == mx[k]);
      const unsigned long long lm = __ballot(i < L && tg[k] < 1.0f);
      if (lm && first_lim == L) first_lim = 64 * k + (int)__builtin_ctzll(lm);
    }
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
  } else {
    /* the reference's per-sample walk (peak_limiter.c:229-249) */
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
      const float g = xl_target_gain(cur_max);
      s_g[A + i] = g;
      if (g < 1.0f && first_lim == L) first_lim = i;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int i = lane + 64 * k;
      tg[k] = i < L ? s_g[A + i] : 1.0f;
    }
    XL_T(4);
  }

  /* the window the next frame starts from */
  const int end_cir = (cir0 + L) % A;
  for (int k = lane; k < A; k += 64) {
    int pos = end_cir + k; /* W[L + k] is the k-th oldest of the new window */
    pos = pos >= A ? pos - A : pos;
    st->max_buf[pos] = s_w[L + k];
  }
  if (lane == 0) {
    const int pos = end_cir + (cur - L);
    st->max_idx = pos >= A ? pos - A : pos;
    st->cir_buf_pnt = end_cir;
  }

  /* ---- 4. finished, or over to the recursion ---- */
  const bool released = psg0 == 1.0; /* with target gain 1 a fixed point of the smoothing; a step there leaves gain_modified = 1 */
  if (released && first_lim >= L) {
    if (lane == 0) {
      p.ws_flag[2 * s] = 1;
      if (L > 0) st->gain_modified = 1.0f;
    }
    apply_frame<CT>(p, s, lane, true, [](int) { return 1.0f; });
    XL_T(5);
    return;
  }
  float *row = p.ws_gain + (int64_t)s * 1024;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int i = lane + 64 * k;
    if (i < L) row[i] = tg[k];
  }
  if (lane == 0) {
    p.ws_flag[2 * s] = 0;
    p.ws_flag[2 * s + 1] = released ? first_lim : 0; /* where the recursion starts */
  }
}

/* ---- step 5: lane = stream ---- */
__global__ __launch_bounds__(64) void xaac_limiter_gain_kernel(XaacLimiterParams p) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  const bool mine = s < p.n_streams && p.ws_flag[2 * (s < p.n_streams ? s : 0)] == 0;
  if (!__ballot(mine)) return;
  const int L = p.frame_len;
  xaac_limiter_state *st = p.state + (mine ? s : 0);
  float *row = p.ws_gain + (int64_t)(mine ? s : 0) * 1024;
  const int start = mine ? p.ws_flag[2 * s + 1] : L;
  XlGain g = {st->gain_modified, st->pre_smoothed_gain};
  if (start > 0) g.gain_modified = 1.0f; /* the released steps before `start` */
  const float ac = st->attack_constant, rc = st->release_constant;
  int first = start;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int v = __shfl_xor(first, o);
    first = v < first ? v : first;
  }
  float4 nxt[4];
  int tile = first >> 4;
  if (mine)
#pragma unroll
    for (int v = 0; v < 4; v++) nxt[v] = reinterpret_cast<const float4 *>(row + 16 * tile)[v];
  for (; 16 * tile < L; tile++) {
    float c[16];
#pragma unroll
    for (int v = 0; v < 4; v++) {
      c[4 * v] = nxt[v].x;
      c[4 * v + 1] = nxt[v].y;
      c[4 * v + 2] = nxt[v].z;
      c[4 * v + 3] = nxt[v].w;
    }
    if (mine && 16 * (tile + 1) < L)
#pragma unroll
      for (int v = 0; v < 4; v++) nxt[v] = reinterpret_cast<const float4 *>(row + 16 * (tile + 1))[v];
    if (!__ballot(mine && start > 16 * tile) && 16 * tile + 16 <= L) {
      /* every stream of the wave is inside its recursion for the whole tile (idle lanes compute along) */
#pragma unroll
      for (int u = 0; u < 16; u++) c[u] = xl_gain_step(g, c[u], ac, rc);
    } else if (mine) {
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int i = 16 * tile + u;
        if (i >= start && i < L) c[u] = xl_gain_step(g, c[u], ac, rc);
      }
    }
    if (mine)
#pragma unroll
      for (int v = 0; v < 4; v++)
        reinterpret_cast<float4 *>(row + 16 * tile)[v] = make_float4(c[4 * v], c[4 * v + 1], c[4 * v + 2], c[4 * v + 3]);
  }
  if (mine) {
    st->gain_modified = g.gain_modified;
    st->pre_smoothed_gain = g.pre_smoothed_gain;
  }
}

template <int CT>
__global__ __launch_bounds__(64) void xaac_limiter_apply_kernel(XaacLimiterParams p) {
  const int lane = threadIdx.x, s = blockIdx.x;
  if (p.ws_flag[2 * s] != 0) return; /* finished by the front kernel */
  const float *row = p.ws_gain + (int64_t)s * 1024;
  apply_frame<CT>(p, s, lane, true, [row](int i) { return row[i]; });
}

extern "C" hipError_t xaac_launch_limiter(const XaacLimiterParams *p, hipStream_t stream) {
  const dim3 grid(p->n_streams), block(64);
  if (p->num_channels == 1)
    hipLaunchKernelGGL(xaac_limiter_front_kernel<1>, grid, block, 0, stream, *p);
  else if (p->num_channels == 2)
    hipLaunchKernelGGL(xaac_limiter_front_kernel<2>, grid, block, 0, stream, *p);
  else
    hipLaunchKernelGGL(xaac_limiter_front_kernel<0>, grid, block, 0, stream, *p);
  hipLaunchKernelGGL(xaac_limiter_gain_kernel, dim3((p->n_streams + 63) / 64), block, 0, stream, *p);
  if (p->num_channels == 1)
    hipLaunchKernelGGL(xaac_limiter_apply_kernel<1>, grid, block, 0, stream, *p);
  else if (p->num_channels == 2)
    hipLaunchKernelGGL(xaac_limiter_apply_kernel<2>, grid, block, 0, stream, *p);
  else
    hipLaunchKernelGGL(xaac_limiter_apply_kernel<0>, grid, block, 0, stream, *p);
  return hipGetLastError();
}

/* xaac_warm_up (xaac_abi.cpp): asking for a kernel's attributes puts this translation unit's code object on the device */
extern "C" hipError_t xaac_warm_limiter(void) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&xaac_limiter_gain_kernel));
}
