/*
 * sbr_core.h -- the "control and adjustment" part of the fixed-point SBR decoder for ONE
 * channel-frame, low-power (real-valued) mode: block-floating-point bookkeeping, LPP transposer
 * (HF generation) and envelope adjustment, as host/device code.  On the GPU one WAVE runs one
 * channel-frame (see "execution context" below); on the host the same code, executed sequentially,
 * is the oracle's arithmetic (oracle/oracle_sbr.cpp), which is pinned to the compiled reference on
 * frames captured from real HE-AAC streams and on fuzzed side info.
 *
 * Reference map (decoder/...):
 *   xs_fix_mant_div / xs_mant_exp_sqrt / xs_fix_div      ixheaacd_basic_funcs.c:66 / :101 / :130
 *   xs_headroom / xs_adjust                               ixheaacd_env_calc.c:1159 / :1099
 *   xs_rescale_x_overlap                                  ixheaacd_sbrdec_lpfuncs.c:453
 *   xs_invfilt_level_emphasis                             ixheaacd_sbrdec_lpfuncs.c:735
 *   xs_low_pow_hf_generator (with the autocorrelation and the patch filter)   ixheaacd_lpp_tran.c:843 (:271, :629 / :665)
 *   xs_map_sineflags                                      ixheaacd_sbrdec_lpfuncs.c:529
 *   xs_energy_per_subband / xs_energy_per_sfb             ixheaacd_env_calc.c:1211 / :1298
 *   xs_subbandgain / xs_calc_subband_gains                ixheaacd_env_calc.c:1382 / :616
 *   xs_avggain / xs_noiselimiting / xs_alias_reduction    ixheaacd_env_calc.c:1454 / :229 / :78
 *   xs_erg_to_amplitude_lp                                ixheaacd_env_calc.c:423
 *   xs_harm_zerotwo_lp / xs_harm_onethree_lp              ixheaacd_env_calc.c:1564 / :1617
 *   xs_adapt_noise_gain / xs_calc_sbrenvelope             ixheaacd_env_calc.c:479 / :692
 *   xs_sbr_core_lp                                        ixheaacd_sbr_dec.c:726-775, :1050-1245, :1283-1308
 * Preconditions the reference's header parser guarantees (ixheaacd_freq_sca.c): band tables strictly
 * increasing, sub_band_end <= 64, at most 56 adjusted bands, patch destinations disjoint and above
 * the source range.
 * The reference keeps gains/energies as interleaved (mantissa, exponent) WORD16 pairs and
 * truncates through WORD16 assignments in many places; the pairs and every truncation are kept.
 */
#ifndef XAAC_SBR_CORE_H
#define XAAC_SBR_CORE_H

#include "fx.h"
#include "../../include/xaac_sbr.h"

#ifndef XS_TABLES_DECLARED
#define XS_TABLES_DECLARED
#if defined(__HIPCC__)
#define XAAC_TAB_QUAL static __device__ const
#include "tables_sbr.inc"
#undef XAAC_TAB_QUAL
#else
#include "tables_sbr.inc"
#endif
#endif

#define XS_MAXF XAAC_SBR_MAX_FREQ_COEFFS

/* x86-style shifts (count taken mod 32), which is what the reference's plain << and >> on int compile to */
FX_HD int32_t xs_shl(int32_t v, int s) { return (int32_t)((uint32_t)v << (s & 31)); }
FX_HD int32_t xs_sar(int32_t v, int s) { return v >> (s & 31); }
FX_HD int xs_pnorm32(int32_t a) { return fx_norm32(a); } /* non-negative arguments only */
FX_HD int16_t xs_mult16_shl_sat(int16_t a, int16_t b) { return fx_sat16(((int32_t)a * b) >> 15); }
FX_HD int16_t xs_mult16_shl(int16_t a, int16_t b) { return (int16_t)(((int32_t)a * b) >> 15); }
FX_HD int16_t xs_mult16(int16_t a, int16_t b) { return (int16_t)(((int32_t)a * b) >> 16); }
FX_HD int32_t xs_mult16x16_shl(int16_t a, int16_t b) { return fx_shl((int32_t)a * b, 1); }
FX_HD int32_t xs_mac16x16_shl_sat(int32_t acc, int16_t b, int16_t c) {
  int32_t p = (int32_t)b * c;
  p = (p != (int32_t)0x40000000) ? fx_shl(p, 1) : FX_MAX32;
  return fx_add_sat(acc, p);
}
FX_HD int16_t xs_shl16_sat(int16_t a, int s) {
  if (s > 15) s = 15;
  return fx_sat16(xs_shl(a, s));
}
/* mult32x16hin32: a * (b >> 16) >> 16 */
FX_HD int32_t xs_mul_hi16(int32_t a, int32_t b) { return fx_mul32x16(a, (int16_t)(b >> 16)); }
FX_HD int32_t xs_mul32x16_shl_sat(int32_t a, int16_t b) {
  if (a == FX_MIN32 && b == (int16_t)-32768) return FX_MAX32;
  return fx_mul32x16_shl(a, b);
}
/* basic_ops.h:100-112 shr32_dir_sat_limit */
FX_HD int32_t xs_shr_dir_sat_limit(int32_t a, int b) {
  if (b < 0) return fx_shl_sat(a, -b);
  return fx_shr(a, b > 31 ? 31 : b);
}

/* ---- pseudo-float helpers (ixheaacd_basic_funcs.c) ------------------------------------------- */
/* The reciprocal and square-root tables sit on the serial paths of the envelope adjuster (one dependent lookup per
   pseudo-float divide); the GPU core kernel points these at its LDS copies before including this file. */
#ifndef XS_TAB_INV
#define XS_TAB_INV(i) xaac_sbr_inv_table[i]
#define XS_TAB_SQRT(i) xaac_sbr_sqrt_table[i]
#endif
/* the four small tables (limiter gains, smoothing filter, 1 / n, chirp targets): the GPU core kernel points these at LDS copies
   too -- as global tables each lookup was a memory latency in the middle of a serial stretch */
#ifndef XS_TAB_LIMG
#define XS_TAB_LIMG(i) xaac_sbr_lim_gains_m[i]
#define XS_TAB_SMOOTH(i) xaac_sbr_smooth_filter[i]
#define XS_TAB_INVINT(i) xaac_sbr_inv_int_table[i]
#define XS_TAB_NEWBW(i) xaac_sbr_new_bw_table[i]
#endif
#ifndef XS_TAB_RAND
#define XS_TAB_RAND(i) xaac_sbr_rand_ph[i] /* the HQ slot loop's complex random phases (the GPU core kernel: an LDS copy) */
#endif
FX_HD int xs_fix_mant_div(int16_t op1, int16_t op2, int16_t *res) {
  int pre = fx_norm32(op2) - 16, post;
  int idx = xs_sar(xs_shl(op2, pre), 16 - 3 - 8) & 511;
  if (idx == 0) {
    post = fx_norm32(op1) - 16;
    *res = (int16_t)xs_shl(op1, post);
  } else {
    idx = (idx - 1) >> 1;
    int32_t ratio = (int32_t)XS_TAB_INV(idx) * op1;
    post = fx_norm32(ratio) - 1;
    *res = (int16_t)(xs_shl(ratio, post) >> 15);
  }
  return pre - post;
}

FX_HD void xs_mant_exp_sqrt(int16_t *me) {
  int32_t m = me[0], e = me[1], rm, re;
  if (m > 0) {
    int pre = fx_norm32((int16_t)m) - 16;
    e -= pre;
    int idx = xs_sar(xs_shl(m, pre), 16 - 3 - 8) & 511;
    rm = XS_TAB_SQRT(idx >> 1);
    if (e & 1) {
      rm = (rm * 0x5a82) >> 16;
      e += 3;
    }
    re = e >> 1;
  } else {
    rm = 0;
    re = -16;
  }
  me[0] = (int16_t)rm;
  me[1] = (int16_t)re;
}

FX_HD int32_t xs_fix_div(int32_t op1, int32_t op2) {
  int32_t q = 0;
  int32_t n1 = op1 >> 1, d1 = op2 >> 1;
  uint32_t num = (uint32_t)(n1 < 0 ? -n1 : n1), den = (uint32_t)(d1 < 0 ? -d1 : d1);
  if (num != 0) {
    for (int k = 15; k > 0; k--) {
      q <<= 1;
      num <<= 1;
      if (num >= den) {
        num -= den;
        q++;
      }
    }
  }
  return ((op1 ^ op2) < 0) ? -q : q;
}

/* accumulate (m, e) into a running (am, ae) pseudo-float sum: the recurring idiom of env_calc.c
   (if e >= ae the sum is shifted down to the new exponent, else the addend is).  Written without the
   branch: shifting by 0 is the identity. */
FX_HD void xs_acc_me(int32_t *am, int32_t *ae, int32_t m, int32_t e) {
  const int32_t d = e - *ae;
  const int32_t up = d > 0 ? d : 0, down = d < 0 ? -d : 0;
  *am = fx_shr(m, down) + fx_shr(*am, up);
  *ae = d >= 0 ? e : *ae;
}

/* ---- execution context ------------------------------------------------------------------------------
 * The same source runs two ways.  On the host (the oracle) lane = 0, n = 1: every lane loop is an
 * ordinary loop and the code is a sequential restatement of the reference.  On the GPU ONE WAVE runs
 * one channel-frame and is used as what it is -- a scalar processor with a 64-wide vector unit:
 *   - XS_PAR / XS_LANES loops spread independent iterations (QMF bands) over the lanes;
 *   - per-band quantities of the envelope adjuster (energies, gains, noise and sine levels, flags)
 *     live in lane vectors (XsLv: element k in lane k, i.e. one VGPR), not in memory;
 *   - the sequential parts of the algorithm (pseudo-float sums over a limiter band, the walks that
 *     build alias groups / per-band metadata / aliasing degrees) run uniformly on all lanes, reading
 *     lane-vector elements with v_readlane -- scalar-ALU code without memory latency;
 *   - the QMF matrix, the state copy and the side info are in LDS, cx.sync() between a producer and a
 *     consumer on another lane.
 * cx.uni() marks a value as wave-uniform (readfirstlane) so loop control stays on the scalar unit. */
struct XsCx {
  int lane, n;
  FX_MEMBER void sync() const {
#if defined(__HIP_DEVICE_COMPILE__)
#if defined(XS_SYNC_WAVE_LDS)
    /* the fixed-point core kernel: a channel-frame is ONE wave and everything it shares between lanes is in LDS, whose
       operations a wave sees in program order -- so producer and consumer only have to stay in order in the compiler.
       No s_barrier (the workgroup's other waves run other channel-frames) and, unlike __syncthreads(), no wait for the
       global loads that are deliberately in flight across these points */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
    __syncthreads();
#endif
#endif
  }
  /* wave reductions of idempotent operations, in the VALU's data-parallel-primitive lanes (no LDS round trips):
     butterfly inside each row of 16, then the row results ripple up; lane 63 ends with the total.  A lane without
     a source keeps its own value, which idempotence makes harmless. */
#if defined(__HIP_DEVICE_COMPILE__)
#define XS_DPP_REDUCE(OP)                                                            \
  {                                                                                  \
    int32_t t_;                                                                      \
    t_ = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false); v = OP;  /* quad_perm [1,0,3,2] */ \
    t_ = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false); v = OP;  /* quad_perm [2,3,0,1] */ \
    t_ = __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false); v = OP; /* row_half_mirror */      \
    t_ = __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false); v = OP; /* row_mirror */           \
    t_ = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false); v = OP; /* row_bcast:15 */         \
    t_ = __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false); v = OP; /* row_bcast:31 */         \
    v = __builtin_amdgcn_readlane(v, 63);                                            \
  }
#endif
  FX_MEMBER int32_t wave_or(int32_t v) const {
#if defined(__HIP_DEVICE_COMPILE__)
    XS_DPP_REDUCE(v | t_)
#endif
    return v;
  }
  FX_MEMBER int32_t wave_max(int32_t v) const {
#if defined(__HIP_DEVICE_COMPILE__)
    XS_DPP_REDUCE(t_ > v ? t_ : v)
#endif
    return v;
  }
  FX_MEMBER int32_t uni(int32_t v) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
  }
  /* first index >= b0 owned by this lane (index k is owned by lane k; ranges end at <= 64) */
  FX_MEMBER int first(int b0) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return lane >= b0 ? lane : lane + 64;
#else
    return b0;
#endif
  }
};
#ifndef XS_T
#define XS_T(i) /* optional phase timer hook (tools/prof_sbr_core.py) */
#endif
#define XS_PAR(k, b0, b1) for (int k = (b0) + cx.lane; k < (b1); k += cx.n)
#define XS_LANES(k, b0, b1) for (int k = cx.first(b0); k < (b1); k += cx.n) /* k on lane k; b1 <= 64 */
#define XS_ONE if (cx.lane == 0)
#ifndef XS_SLOT_UNROLL
#define XS_SLOT_UNROLL 4 /* slots whose LDS reads are in flight together in the column walks */
#endif
/* XS_KEEP(v): the value is what it is, here (device: an empty asm the optimiser cannot look through -- keeps a batch of loads
   a batch; host: nothing) */
#if defined(__HIP_DEVICE_COMPILE__)
#define XS_KEEP(v) asm volatile("" : "+v"(v))
#else
#define XS_KEEP(v)
#endif
#define XS_PRAGMA_(x) _Pragma(#x)
#define XS_PRAGMA(x) XS_PRAGMA_(x)
#if defined(__HIPCC__) && !defined(XS_NO_UNROLL)
#define XS_UNROLL4 XS_PRAGMA(unroll XS_SLOT_UNROLL)
#define XS_UNROLL8 _Pragma("unroll 8")
#define XS_UNROLL _Pragma("unroll")
#elif defined(__HIPCC__)
#define XS_UNROLL4 _Pragma("nounroll")
#define XS_UNROLL8 _Pragma("unroll 8")
#define XS_UNROLL _Pragma("unroll")
#else
#define XS_UNROLL4
#define XS_UNROLL8
#define XS_UNROLL
#endif

/* Lane vector: 64 int32 elements, element k held by lane k (a VGPR) / a plain array in the oracle. */
struct XsLv {
#if defined(__HIP_DEVICE_COMPILE__)
  int32_t v;
  FX_MEMBER int32_t &own(int) { return v; }
  FX_MEMBER int32_t own(int) const { return v; }
  FX_MEMBER int32_t get(int i) const { return __builtin_amdgcn_readlane(v, i); } /* i uniform */
  FX_MEMBER void put(int i, int32_t val) { /* i, val uniform */
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    v = lane == i ? val : v;
  }
  FX_MEMBER void fill(int32_t val) { v = val; }
  FX_MEMBER XsLv shifted(const XsCx &cx, int d) const { /* element k of the result = element k + d (0 outside) */
    XsLv r;
    const int s = cx.lane + d;
    const int32_t t = __shfl(v, s & 63);
    r.v = (s >= 0 && s < 64) ? t : 0;
    return r;
  }
  /* element k of the result = wrapping sum of the elements congruent to k mod w (w = 16, 32 or 64) */
  FX_MEMBER XsLv fold(int w) const {
    XsLv r;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    int32_t t = v;
    if (w <= 32) t = (int32_t)((uint32_t)t + (uint32_t)__shfl(t, (lane + 32) & 63));
    if (w <= 16) t = (int32_t)((uint32_t)t + (uint32_t)__shfl(t, (lane + 16) & 63));
    r.v = t;
    return r;
  }
  FX_MEMBER XsLv gather(const XsLv &idx) const { /* element k of the result = element idx[k] (all lanes take part) */
    XsLv r;
    r.v = __shfl(v, idx.v & 63);
    return r;
  }
#else
  int32_t a[64];
  FX_MEMBER int32_t &own(int k) { return a[k & 63]; }
  FX_MEMBER int32_t own(int k) const { return a[k & 63]; }
  FX_MEMBER int32_t get(int i) const { return a[i & 63]; }
  FX_MEMBER void put(int i, int32_t val) { a[i & 63] = val; }
  FX_MEMBER void fill(int32_t val) {
    for (int i = 0; i < 64; i++) a[i] = val;
  }
  FX_MEMBER XsLv shifted(const XsCx &, int d) const {
    XsLv r;
    for (int i = 0; i < 64; i++) r.a[i] = (i + d >= 0 && i + d < 64) ? a[i + d] : 0;
    return r;
  }
  FX_MEMBER XsLv gather(const XsLv &idx) const {
    XsLv r;
    for (int i = 0; i < 64; i++) r.a[i] = a[idx.a[i] & 63];
    return r;
  }
  FX_MEMBER XsLv fold(int w) const {
    XsLv r;
    for (int i = 0; i < 64; i++) {
      uint32_t t = 0;
      for (int j = i % w; j < 64; j += w) t += (uint32_t)a[j];
      r.a[i] = (int32_t)t;
    }
    return r;
  }
#endif
};

/* OR of all 64 elements */
FX_HD int32_t xs_lv_or(const XsCx &cx, const XsLv &v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return cx.wave_or(v.v);
#else
  int32_t m = 0;
  for (int i = 0; i < 64; i++) m |= v.a[i];
  (void)cx;
  return m;
#endif
}

/* (mantissa, exponent) pseudo-float packed into one lane-vector element */
FX_HD int32_t xs_me(int16_t m, int16_t e) { return (int32_t)(((uint32_t)(uint16_t)e << 16) | (uint16_t)m); }
FX_HD int16_t xs_m(int32_t v) { return (int16_t)v; }
FX_HD int16_t xs_e(int32_t v) { return (int16_t)(v >> 16); }

/* number of elements 0..k (inclusive) of s[0..n) whose mantissa is non-zero */
FX_HD XsLv xs_prefix_nonzero_m(const XsCx &cx, const XsLv &s, int n) {
  XsLv r;
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long mask = __ballot(cx.lane < n && xs_m(s.v) != 0);
  r.v = __popcll(mask & ((2ull << cx.lane) - 1ull));
#else
  int c = 0;
  for (int k = 0; k < 64; k++) {
    if (k < n && xs_m(s.a[k]) != 0) c++;
    r.a[k] = c;
  }
#endif
  return r;
}

/* bit `bit` of elements 0..n-1 gathered into a 64-bit mask */
FX_HD uint64_t xs_ballot(const XsCx &cx, const XsLv &p, int bit, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __ballot(cx.lane < n && (p.v & bit) != 0);
#else
  uint64_t m = 0;
  for (int k = 0; k < 64 && k < n; k++)
    if (p.a[k] & bit) m |= (uint64_t)1 << k;
  return m;
#endif
}
FX_HD int xs_clz64(uint64_t v) { return __builtin_clzll(v); } /* v != 0 */
FX_HD int xs_ctz64(uint64_t v) { return __builtin_ctzll(v); } /* v != 0 */
FX_HD uint64_t xs_mask_upto(int k) { return ((uint64_t)2 << k) - 1; } /* bits 0..k, k <= 63 */

/* ---- the order-dependent pseudo-float sums, split into what is a recursion and what is not ------------------
 * xs_acc_me over a sequence (m_1, e_1), (m_2, e_2), ... from (0, 0) leaves ae = E_n and am = F_n with
 *     E_k = max(0, e_1 .. e_k),      F_k = (F_{k-1} >> (E_k - E_{k-1})) + (m_k >> (E_k - e_k)).
 * The running exponents E_k are a prefix maximum -- associative, so a scan over the lanes --, the addends
 * c_k = m_k >> (E_k - e_k) and the shifts d_k = E_k - E_{k-1} follow per element in parallel, and only
 * F_k = (F_{k-1} >> d_k) + c_k stays sequential: two dependent operations per element instead of the dozen of
 * the compare / select / shift / add form.  Exact: it is the same arithmetic in the same order. */

/* inclusive prefix maximum over the lanes: element k = max of elements 0..k (elements >= 0) */
FX_HD XsLv xs_prefix_max(const XsCx &cx, const XsLv &s) {
  XsLv r = s;
#if defined(__HIP_DEVICE_COMPILE__)
  /* elements are >= 0 (0 = nothing): the data-parallel-primitive shifts feed 0 into the lanes without a source.
     Inside each row of 16 by shifts of 1, 2, 4, 8; then every row takes the last element of the rows below it. */
#define XS_DPP_MAX(ctrl, rows)                                                      \
  {                                                                                 \
    const int32_t t_ = __builtin_amdgcn_update_dpp(0, r.v, ctrl, rows, 0xf, false); \
    r.v = t_ > r.v ? t_ : r.v;                                                      \
  }
  XS_DPP_MAX(0x111, 0xf) /* row_shr:1 */
  XS_DPP_MAX(0x112, 0xf) /* row_shr:2 */
  XS_DPP_MAX(0x114, 0xf) /* row_shr:4 */
  XS_DPP_MAX(0x118, 0xf) /* row_shr:8 */
  XS_DPP_MAX(0x142, 0xa) /* row_bcast:15 into rows 1 and 3 */
  XS_DPP_MAX(0x143, 0xc) /* row_bcast:31 into rows 2 and 3 */
#undef XS_DPP_MAX
  (void)cx;
#else
  for (int k = 1; k < 64; k++) r.a[k] = r.a[k - 1] > r.a[k] ? r.a[k - 1] : r.a[k];
#endif
  return r;
}

/* E before element k: max(0, e_j) over the elements j < k of k's segment.  seg: segment of each element, non-decreasing
   along the lanes and below 2^10, -1 = not in a sum.  |e| < 2^19: any sum of two 16-bit exponents fits. */
FX_HD XsLv xs_seg_running_max(const XsCx &cx, const XsLv &seg, const XsLv &e, int n) {
  XsLv key;
  key.fill(0);
  XS_LANES(k, 0, n) key.own(k) = seg.own(k) >= 0 ? ((seg.own(k) + 1) << 20) | ((e.own(k) + 0x80000) & 0xfffff) : 0;
  const XsLv before = xs_prefix_max(cx, key).shifted(cx, -1);
  XsLv r;
  r.fill(0);
  XS_LANES(k, 0, n) {
    const int32_t p = before.own(k);
    const int32_t ep = (p & 0xfffff) - 0x80000;
    r.own(k) = ((p >> 20) == seg.own(k) + 1 && ep > 0) ? ep : 0;
  }
  return r;
}

/* ---- QMF matrix view: rows -2,-1 are the LPC history, rows 0..37 the slots.  Low-power mode keeps 64
   real values per slot; HQ mode keeps 64 real then 64 imaginary ones (the reference's slot-pointer
   arrays over one scratch block, sbr_dec.c:752-766, have exactly these strides). */
/* NB_: bands a row holds.  64 is the reference's layout; the GPU core kernel keeps rows of NB_ < 64 bands (less LDS,
   more resident waves) for streams that provably touch no band at or above NB_ and sends the others through the
   64-band instantiation (sbr_core_kernel.hip: "narrow rows"). */
/* LD_: the low-delay SBR of AAC-ELD (sbr_dec.c:706-775: op_delay 0 -- no overlap slots, nothing carried in the matrix --, one QMF
   slot per time slot, 16 or 15 slots a frame): rows -2, -1 the LPC history, rows 0 .. cols - 1 the frame's slots. */
template <int HQ_, int NB_ = 64, int LD_ = 0>
struct XsQmfT {
  static constexpr int HQ = HQ_;
  static constexpr int NB = NB_;
  static constexpr int LD = LD_;
  static constexpr int OV = LD_ ? 0 : 6; /* op_delay: the overlap slots in front of the frame's own */
  static constexpr int IM = NB_;                    /* offset of a row's imaginary columns */
  static constexpr int ROW = HQ_ ? 2 * NB_ : NB_;
  int32_t *p;
  FX_MEMBER int32_t &operator()(int slot, int band) const { return p[(slot + 2) * ROW + band]; }
  FX_MEMBER int32_t &im(int slot, int band) const { return p[(slot + 2) * ROW + IM + band]; }
};
typedef XsQmfT<0> XsQmf;
typedef XsQmfT<1> XsQmfHq;

struct XsCov {
  int32_t phi_11, phi_22, phi_01, phi_02, phi_12, d;
};

/* Per-channel-frame scratch in memory shared by the lanes (LDS on the GPU, stack in the oracle). */
struct XsWork {
  int32_t bw_array[XAAC_SBR_MAX_PATCHES];
  union { /* never live together: nrg_est is consumed into the est lane vector before the alias reduction starts */
    int32_t fold_a[XS_MAXF + 8][2]; /* inputs of the sequential pseudo-float sums, per band */
    int16_t nrg_est[2 * XS_MAXF];   /* only the interpol_freq == 0 path goes through memory */
  };
  int32_t fold_b[XS_MAXF + 8][4];
  int16_t res_a[XS_MAXF + 8][4];  /* their results, per limiter band / per alias group (by first band) */
  int16_t res_b[XS_MAXF + 8][2];
};

/* Per-band registers of the envelope adjuster.  est / e_orig / gain / noise / sine / meta: element c
   is QMF band max_qmf_subband_aac + c; alias_red / sine_mapped / deg / deg1: element i is band
   sub_band_start + i (deg1 is deg one band up).  Exactly the index conventions of the reference's
   nrg_est[] ... arrays (env_calc.c:692), two int16 per entry there, one packed int32 here. */
struct XsEnv {
  XsLv est, e_orig, gain, noise, sine, meta, alias_red, sine_mapped, deg, deg1;
};

/* env_calc.c:1159: headroom of bands [b0,b1) x slots [s0,s1).  In a complex matrix the real and the imaginary
   columns of the range are spread over the lanes side by side (a range of up to 32 bands fills the wave). */
template <class Q>
FX_HD int xs_headroom(const XsCx &cx, const Q &x, int b0, int b1, int s0, int s1) {
  int32_t m = 1;
  const int nb = b1 > b0 ? b1 - b0 : 0;
  XS_PAR(c, 0, Q::HQ ? 2 * nb : nb) {
    const int col = c < nb ? b0 + c : Q::IM + b0 + (c - nb);
    XS_UNROLL4
    for (int l = s0; l < s1; l++) m |= fx_abs_nrm(x(l, col));
  }
  return xs_pnorm32(cx.wave_or(m));
}
/* the same, sequential (for code that already runs one band group per lane) */
template <class Q>
FX_HD int xs_headroom_seq(const Q &x, int b0, int b1, int s0, int s1) {
  int32_t m = 1;
  for (int l = s0; l < s1; l++)
    for (int k = b0; k < b1; k++) {
      m |= fx_abs_nrm(x(l, k));
      if (Q::HQ) m |= fx_abs_nrm(x.im(l, k));
    }
  return xs_pnorm32(m);
}
/* env_calc.c:1099 */
template <class Q>
FX_HD void xs_adjust(const XsCx &cx, const Q &x, int b0, int b1, int s0, int s1, int shift) {
  if (shift == 0) return;
  if (shift > 31) shift = 31;
  if (shift < -31) shift = -31;
  const int nb = b1 > b0 ? b1 - b0 : 0;
  XS_PAR(c, 0, Q::HQ ? 2 * nb : nb) {
    const int col = c < nb ? b0 + c : Q::IM + b0 + (c - nb); /* real and imaginary columns side by side */
    XS_UNROLL4
    for (int l = s0; l < s1; l++) x(l, col) = shift > 0 ? fx_shlw(x(l, col), shift) : (x(l, col) >> -shift);
  }
}

/* zero bands [b0,b1) x slots [s0,s1) (real and imaginary parts) */
template <class Q>
FX_HD void xs_clear(const XsCx &cx, const Q &x, int b0, int b1, int s0, int s1) {
  const int nb = b1 > b0 ? b1 - b0 : 0;
  XS_PAR(c, 0, Q::HQ ? 2 * nb : nb) {
    const int col = c < nb ? b0 + c : Q::IM + b0 + (c - nb);
    for (int l = s0; l < s1; l++) x(l, col) = 0;
  }
}
/* sbr_dec.c:1221-1236: the last two slots of the low bands are the next frame's LPC history */
template <class ST, class Q>
FX_HD void xs_lpc_save(const XsCx &cx, ST *st, const Q &x, int usb, int no_bins = 32) {
  XS_PAR(k, 0, usb) {
    st->lpc_real[0][k] = x(no_bins - 2, k);
    st->lpc_real[1][k] = x(no_bins - 1, k);
    if (Q::HQ) {
      st->lpc_imag[0][k] = x.im(no_bins - 2, k);
      st->lpc_imag[1][k] = x.im(no_bins - 1, k);
    }
  }
}

/* ---- HF generator, low-power mode ------------------------------------------------------------------ */
/* sbrdec_lpfuncs.c:735 (at most 5 inverse-filtering bands: one per lane) */
FX_HD void xs_invfilt_level_emphasis(const XsCx &cx, const int32_t *bw_prev, int n, const int32_t *mode,
                                     const int32_t *mode_prev, int32_t *bw) {
  XS_PAR(i, 0, n) {
    int32_t b = XS_TAB_NEWBW(4 * mode_prev[i] + mode[i]);
    int16_t w1, w2;
    if (b < bw_prev[i]) {
      w1 = 0x6000;
      w2 = 0x2000;
    } else {
      w1 = 0x7400;
      w2 = 0x0c00;
    }
    int32_t a = fx_add(fx_mul32x16_shl(b, w1), fx_mul32x16_shl(bw_prev[i], w2));
    if (a < 0x02000000) a = 0;
    if (a >= 0x7f800000) a = 0x7f800000;
    bw[i] = a;
  }
}

/* lpp_tran.c:665, first half: prediction coefficients and reflection coefficient of one low band */
FX_HD void xs_lpc_coeffs_lp(const XsCov *c, int16_t *alpha0_out, int16_t *alpha1_out, int16_t *k1_out) {
  int16_t alpha0 = 0, alpha1 = 0, k1;
  if (c->d != 0) {
    int norm_d = fx_norm32(c->d);
    int16_t inv_d = (int16_t)xs_fix_div(0x40000000, xs_shl(c->d, norm_d));
    int32_t mod_d = c->d < 0 ? -c->d : c->d;
    int32_t t = fx_sub_sat(fx_mul32(c->phi_01, c->phi_12), fx_mul32(c->phi_02, c->phi_11)) >> 2;
    if ((t < 0 ? -t : t) < mod_d) alpha1 = (int16_t)(xs_shl(xs_mul32x16_shl_sat(t, inv_d), norm_d) >> 15);
    t = fx_sub_sat(fx_mul32(c->phi_02, c->phi_12), fx_mul32(c->phi_01, c->phi_22)) >> 2;
    if ((t < 0 ? -t : t) < mod_d) alpha0 = (int16_t)(xs_shl(xs_mul32x16_shl_sat(t, inv_d), norm_d) >> 15);
  }
  if (c->phi_11 == 0) {
    k1 = 0;
  } else if (fx_abs_sat(c->phi_01) >= c->phi_11) {
    k1 = c->phi_01 < 0 ? (int16_t)0x7fff : (int16_t)-0x8000;
  } else {
    k1 = (int16_t)(-((int16_t)xs_fix_div(c->phi_01, c->phi_11)));
  }
  *alpha0_out = alpha0;
  *alpha1_out = alpha1;
  *k1_out = k1;
}

/* lpp_tran.c:665, aliasing degrees.  The reference's loop over the low bands carries the reflection
   coefficients of the two bands below along and, at band L, writes deg[L] and sometimes overwrites
   deg[L-1].  Spelled out, the step at L only reads k1[L], k1[L-1], k1[L-2] (0 below start_patch), so
   band lb's final value is what step lb+1 wrote over it, else what step lb wrote: one band per lane. */
FX_HD void xs_degree_step(int L, int16_t a, int16_t b, int16_t c, int32_t *here, int32_t *below, int *below_set) {
  const int16_t dg = fx_sat16(0x7fff - (int32_t)xs_mult16_shl_sat(b, b));
  *here = 0;
  *below_set = 0;
  *below = dg;
  if (((L & 1) == 0) && (a < 0)) {
    if (b < 0) {
      *here = 0x7fff;
      if (c > 0) *below_set = 1;
    } else if (c > 0) {
      *here = dg;
    }
  }
  if (((L & 1) != 0) && (a > 0)) {
    if (b > 0) {
      *here = 0x7fff;
      if (c < 0) *below_set = 1;
    } else if (c < 0) {
      *here = dg;
    }
  }
}
FX_HD void xs_degree_alias_lp(const XsCx &cx, const XsLv &k1v, XsLv &deg, int start_patch, int stop_patch) {
  const XsLv km1 = k1v.shifted(cx, -1), km2 = k1v.shifted(cx, -2), kp1 = k1v.shifted(cx, 1);
  XS_LANES(lb, 0, stop_patch) {
    /* k1v is 0 outside [start_patch, stop_patch), which is what the reference's initial values are */
    int32_t here = 0, below = 0, h2 = 0;
    int set = 0, dummy = 0;
    if (lb >= start_patch && lb > 1) {
      xs_degree_step(lb, (int16_t)k1v.own(lb), (int16_t)km1.own(lb), (int16_t)km2.own(lb), &here, &h2, &dummy);
      deg.own(lb) = here;
    }
    const int L = lb + 1;
    if (L >= start_patch && L < stop_patch && L > 1) {
      xs_degree_step(L, (int16_t)kp1.own(lb), (int16_t)k1v.own(lb), (int16_t)km1.own(lb), &h2, &below, &set);
      if (set) deg.own(lb) = below;
    }
  }
}

/* lpp_tran.c:843.  deg (64 aliasing degrees by QMF band) must be zeroed by the caller.  Writes
   bw_array_prev. */
template <class ST, class Q>
FX_HD void xs_low_pow_hf_generator(const XsCx &cx, const xaac_sbr_header *h, ST *st, const Q &x, XsWork *w,
                                   XsLv &deg, int start_idx, int last_slot_offset, int max_qmf_subband,
                                   const int32_t *invf_mode, const int32_t *invf_mode_prev, int norm_max) {
  const int num_patches = cx.uni(h->num_patches);
  const int auto_corr_len = cx.uni(h->num_columns) + 6;
  const int stop_idx = cx.uni(h->num_columns) + last_slot_offset;
  XS_PAR(i, 0, XAAC_SBR_MAX_PATCHES) w->bw_array[i] = 0;
  cx.sync();
  xs_invfilt_level_emphasis(cx, st->bw_array_prev, h->num_if_bands, invf_mode, invf_mode_prev, w->bw_array);
  const int actual_stop = cx.uni(
      (int16_t)(h->patch[num_patches - 1].dst_start_band + h->patch[num_patches - 1].num_bands_in_patch));
  {
    int len = 6;
    if (len > stop_idx) len = stop_idx;
    XS_PAR(k, actual_stop, 64)
      for (int l = start_idx; l <= len - 1; l++) x(l, k) = 0;
    if (actual_stop < 32) XS_PAR(k, actual_stop, 32)
      for (int l = len; l <= stop_idx - 1; l++) x(l, k) = 0;
  }
  int start_patch = cx.uni(h->start_patch) - 2;
  if (start_patch < 1) start_patch = 1;
  const int stop_patch = cx.uni(h->patch[0].dst_start_band);
  XS_PAR(k, 0, stop_patch) {
    x(-2, k) = st->lpc_real[0][k];
    x(-1, k) = st->lpc_real[1][k];
  }
  cx.sync();
  XS_T(12);
  XsLv k1v, alpha;
  k1v.fill(0);
  alpha.fill(0);
  /* lpp_tran.c:271: real autocorrelation of band k over auto_corr_len (= 38) slots from row -2 on (the reference walks
     three samples at a time; the running sums only depend on the sample order).  The sums wrap (fx_add), so their order is free: the slots of a band are split over as many
     lane groups as the low bands leave room for (W lanes per group: lane l takes the band congruent to l modulo W and
     the l / W-th part of the slots) and the parts are added up across the groups; lane k then finishes band k. */
  const int nlow = stop_patch - start_patch;
  const int W = nlow <= 16 ? 16 : nlow <= 32 ? 32 : 64, G = 64 / W;
  XsLv p01v, p02v, p11v;
  p01v.fill(0);
  p02v.fill(0);
  p11v.fill(0);
  if (norm_max != 30) {
    XS_LANES(l, 0, 64) {
      const int k = start_patch + ((l - start_patch) & (W - 1)), g = l / W;
      if (k < stop_patch) {
        const int j0 = g * auto_corr_len / G, j1 = (g + 1) * auto_corr_len / G;
        int32_t p01 = 0, p02 = 0, p11 = 0;
        int32_t t1 = fx_shr(x(j0 - 2, k), 3), t2 = fx_shr(x(j0 - 1, k), 3);
        XS_UNROLL4
        for (int j = j0; j < j1; j++) {
          const int32_t t3 = fx_shr(x(j, k), 3);
          p01 = fx_add(p01, xs_mul_hi16(t3, t2));
          p02 = fx_add(p02, xs_mul_hi16(t3, t1));
          p11 = fx_add(p11, xs_mul_hi16(t2, t2));
          t1 = t2;
          t2 = t3;
        }
        p01v.own(l) = p01;
        p02v.own(l) = p02;
        p11v.own(l) = p11;
      }
    }
    if (G > 1) {
      p01v = p01v.fold(W);
      p02v = p02v.fold(W);
      p11v = p11v.fold(W);
    }
  }
  XS_LANES(k, start_patch, stop_patch) {
    XsCov c = {0, 0, 0, 0, 0, 0};
    if (norm_max != 30) {
      /* lpp_tran.c:271 behind its loop: the samples at the two ends of the 38 slots */
      const int32_t p01 = p01v.own(k), p02 = p02v.own(k), p11 = p11v.own(k);
      const int32_t first = fx_shr(x(-2, k), 3), second = fx_shr(x(-1, k), 3);
      const int32_t last1 = fx_shr(x(auto_corr_len - 1, k), 3), last3 = fx_shr(x(auto_corr_len - 2, k), 3);
      const int32_t p12 = fx_add(fx_sub(p01, xs_mul_hi16(last1, last3)), xs_mul_hi16(second, first));
      const int32_t p22 = fx_add(fx_sub(p11, xs_mul_hi16(last3, last3)), xs_mul_hi16(first, first));
      const int32_t mx = fx_abs_nrm(p01) | fx_abs_nrm(p02) | fx_abs_nrm(p12) | p11 | p22;
      const int q = xs_pnorm32(mx);
      c.phi_11 = xs_shl(p11, q);
      c.phi_22 = xs_shl(p22, q);
      c.phi_01 = xs_shl(p01, q);
      c.phi_02 = xs_shl(p02, q);
      c.phi_12 = xs_shl(p12, q);
      c.d = fx_sub_sat(fx_mul32(c.phi_22, c.phi_11), fx_mul32(c.phi_12, c.phi_12));
    }
    int16_t a0, a1, k1;
    xs_lpc_coeffs_lp(&c, &a0, &a1, &k1);
    alpha.own(k) = xs_me(a0, a1);
    k1v.own(k) = k1;
  }
  XS_T(13);
  xs_degree_alias_lp(cx, k1v, deg, start_patch, stop_patch);
  XS_T(14);
  /* lpp_tran.c:629 + :665, second half: low band lb copied / inverse-filtered into each patch's high band.  One lane per
     HIGH band (the reference walks the low bands and, per low band, the patches: fifteen lanes with three patches each in a
     common stream); with at most 32 high bands the two halves of the wave share a band's slots.  A slot of a high band reads
     its low band's slots only, so the split is free; the patches' destinations are disjoint (a later patch would win). */
  {
    const int nhigh = actual_stop - stop_patch;
    const bool halves = nhigh <= 32;
    XsLv lbv, pv;
    lbv.fill(0);
    pv.fill(-1);
    XS_LANES(l, 0, 64) {
      const int hb = halves ? stop_patch + (l & 31) : l;
      if (hb < max_qmf_subband || hb >= 64) continue;
      for (int patch = 0; patch < num_patches; patch++) {
        const xaac_sbr_patch *pp = &h->patch[patch];
        const int lb = hb - pp->dst_end_band;
        if (lb < start_patch || lb >= stop_patch || lb < pp->src_start_band || lb >= pp->src_end_band) continue;
        if ((xs_shl(lb + pp->dst_end_band, 8) >> 8) != hb) continue;
        lbv.own(l) = lb;
        pv.own(l) = patch;
      }
    }
    const XsLv al = alpha.gather(lbv);
    XS_LANES(l, 0, 64) {
      if (pv.own(l) < 0) continue;
      const int hb = halves ? stop_patch + (l & 31) : l, part = halves ? l >> 5 : 0, lb = lbv.own(l);
      const int16_t alpha0 = xs_m(al.own(l)), alpha1 = xs_e(al.own(l));
      int bi = 0;
      /* (the reference's running index, lpp_tran.c:665: the first border above the band -- the last border of a parser's
         table is sub_band_end; stopped where bw_array ends for a table that has none above) */
      while (bi < XAAC_SBR_MAX_PATCHES - 1 && hb >= h->bw_borders[bi]) bi++;
      int16_t bw = (int16_t)(w->bw_array[bi] >> 16);
      const int32_t a0 = xs_mult16x16_shl(bw, alpha0);
      bw = xs_mult16_shl_sat(bw, bw);
      const int32_t a1 = xs_mult16x16_shl(bw, alpha1);
      const int len = stop_idx - start_idx - 1;
      /* the reference produces two slots per step when it filters: an odd count runs one slot past stop_idx */
      const int n_tot = bw > 0 ? (len >= 0 ? ((len >> 1) + 1) * 2 : 0) : len + 1;
      const int n_half = halves ? (n_tot + 1) >> 1 : n_tot;
      const int i0 = part ? n_half : 0, i1 = part ? n_tot : n_half;
      if (bw > 0) {
        /* lpp_tran.c:629: y[n] = x[n]/4 + 2 (a1 x[n-2] + a0 x[n-1]) */
        int32_t xm2 = x(start_idx + i0 - 2, lb), xm1 = x(start_idx + i0 - 1, lb);
        XS_UNROLL4
        for (int i = i0; i < i1; i++) {
          const int32_t cur = x(start_idx + i, lb);
          const int32_t t = xs_mul_hi16(xm2, a1);
          x(start_idx + i, hb) = fx_add_sat(cur >> 2, fx_shlw(fx_add(t, xs_mul_hi16(xm1, a0)), 1));
          xm2 = xm1;
          xm1 = cur;
        }
      } else {
        XS_UNROLL4
        for (int i = i0; i < i1; i++) x(start_idx + i, hb) = x(start_idx + i, lb) >> 2;
      }
    }
  }
  /* lpp_tran.c:905: the high bands inherit the aliasing degree of their source band (patch
     destinations are disjoint and above the source range, so one shifted copy per patch) */
  {
    const int lb0 = cx.uni(h->start_patch), lb1 = cx.uni(h->stop_patch);
    for (int patch = 0; patch < num_patches; patch++) {
      const xaac_sbr_patch *pp = &h->patch[patch];
      const int off = cx.uni(pp->dst_end_band), s0 = cx.uni(pp->src_start_band), s1 = cx.uni(pp->src_end_band);
      const int d0 = cx.uni(pp->dst_start_band);
      const XsLv src = deg.shifted(cx, -off);
      XS_LANES(hb, 0, 64) {
        const int lb = hb - off;
        if (lb >= lb0 && lb < lb1 && lb >= s0 && lb < s1 && hb != d0) deg.own(hb) = src.own(hb);
      }
    }
  }
  XS_PAR(i, 0, h->num_if_bands) st->bw_array_prev[i] = w->bw_array[i];
  cx.sync();
}

/* ---- envelope adjuster ------------------------------------------------------------------------------ */
/* sbrdec_lpfuncs.c:529.  harm_flags_prev[j] pairs with sfb nsf-1-j; two sfbs can map to the same
   band, the lower sfb (visited later by the reference) wins, hence the sequential stores. */
template <class ST>
FX_HD void xs_map_sineflags(const XsCx &cx, const int16_t *tbl_hi, int nsf, const uint8_t *add_harm, ST *st,
                            int tr_env, XsLv &sine_mapped) {
  const int low2 = cx.uni(tbl_hi[0]) << 1;
  XsLv q, val;
  q.fill(-1);
  val.fill(0);
  sine_mapped.fill(XAAC_SBR_MAX_ENVELOPES);
  XS_LANES(i, 0, nsf) {
    const int j = nsf - 1 - i;
    const int old = st->harm_flags_prev[j];
    st->harm_flags_prev[j] = (int8_t)add_harm[i];
    if (add_harm[i]) q.own(i) = ((tbl_hi[i + 1] + tbl_hi[i]) - low2) >> 1;
    val.own(i) = old ? 0 : (int8_t)tr_env;
  }
  XsLv has;
  has.fill(0);
  XS_LANES(i, 0, nsf) has.own(i) = q.own(i) >= 0;
  uint64_t todo = xs_ballot(cx, has, 1, nsf);
  while (todo) { /* descending sfb order, as the reference stores them */
    const int i = 63 - xs_clz64(todo);
    todo &= ~((uint64_t)1 << i);
    sine_mapped.put(q.get(i), val.get(i));
  }
}

/* env_calc.c:1211: energy estimate of band k over slots [s0,s1) (HQ: real and imaginary parts) */
template <class Q>
FX_HD int32_t xs_energy_of_subband(const Q &x, int s0, int s1, int k, int frame_exp2, int16_t inv_width) {
  const int n = s1 - s0;
  int32_t mx = 1;
  XS_UNROLL4
  for (int l = 0; l < n; l++) {
    int32_t v = fx_abs_nrm(x(s0 + l, k));
    if (v > mx) mx = v;
    if (Q::HQ) {
      v = fx_abs_nrm(x.im(s0 + l, k));
      if (v > mx) mx = v;
    }
  }
  int pre = xs_pnorm32(mx) - (Q::HQ ? 4 : 3);
  int32_t accu = 0;
  int shift = 16 - pre;
  /* shift > 0 ? xs_sar(v, shift) : xs_shl(v, -shift) without the branch: one of the two counts is zero */
  const int e_shr = shift > 0 ? shift & 31 : 0, e_shl = shift > 0 ? 0 : (-shift) & 31;
  XS_UNROLL4
  for (int l = 0; l < n; l++) {
    int16_t t = (int16_t)((int32_t)((uint32_t)x(s0 + l, k) << e_shl) >> e_shr);
    accu = fx_add(accu, (int32_t)t * t);
    if (Q::HQ) {
      t = (int16_t)((int32_t)((uint32_t)x.im(s0 + l, k) << e_shl) >> e_shr);
      accu = fx_add(accu, (int32_t)t * t);
    }
  }
  if (accu == 0) return 0;
  shift = -xs_pnorm32(accu);
  int16_t sum_m = (int16_t)xs_shr_dir_sat_limit(accu, 16 + shift);
  sum_m = xs_mult16_shl_sat(sum_m, inv_width);
  shift = shift - (pre << 1) + (Q::HQ ? 0 : 1);
  return xs_me(sum_m, (int16_t)(frame_exp2 + shift + 1));
}
template <class Q>
FX_HD void xs_energy_per_subband(const XsCx &cx, const Q &x, int s0, int s1, int b0, int b1, int frame_exp,
                                 XsLv &est) {
  const int16_t inv_width = XS_TAB_INVINT(s1 > s0 ? s1 - s0 : 0); /* (an envelope with its borders out of order has no slots: the
                                                                     reference reads in front of its table there) */
  XS_LANES(c, 0, b1 - b0) est.own(c) = xs_energy_of_subband(x, s0, s1, b0 + c, frame_exp << 1, inv_width);
}

/* env_calc.c:1298: one scale-factor band per lane.  The reference appends the estimates of successive
   sfbs; they start at the first sfb at or above max_sb. */
template <class Q>
FX_HD void xs_energy_per_sfb(const XsCx &cx, const Q &x, int nsf, const int16_t *tbl, int s0, int s1, int max_sb,
                             int frame_exp, XsWork *w, XsLv &est) {
  const int16_t inv_width = XS_TAB_INVINT(s1 > s0 ? s1 - s0 : 0); /* (an envelope with its borders out of order has no slots: the
                                                                     reference reads in front of its table there) */
  frame_exp <<= 1;
  int j0 = 0;
  while (j0 < nsf && cx.uni(tbl[j0]) < max_sb) j0++;
  const int base = cx.uni(tbl[j0]);
  XS_PAR(j, j0, nsf) {
    const int li = tbl[j], ui = tbl[j + 1];
    int pre = xs_headroom_seq(x, li, ui, s0, s1) - 4;
    int32_t accumulate = 0;
    for (int k = li; k < ui; k++) {
      int p1 = 16 - pre;
      if (p1 > 31) p1 = 31;
      int32_t line = 0;
      for (int l = s0; l < s1; l++) {
        int16_t t = (int16_t)fx_shr_dir(x(l, k), p1);
        line = fx_add_sat(line, (int32_t)t * t);
        if (Q::HQ) {
          t = (int16_t)fx_shr_dir(x.im(l, k), p1);
          line = fx_add_sat(line, (int32_t)t * t);
        }
      }
      accumulate = fx_add_sat(accumulate, fx_shr(line, 9));
    }
    int shift = xs_pnorm32(accumulate);
    int16_t sum_m = (int16_t)xs_shr_dir_sat_limit(accumulate, 16 - shift);
    int32_t sum_e;
    if (sum_m == 0) {
      sum_e = 0;
    } else {
      sum_m = xs_mult16_shl_sat(sum_m, inv_width);
      sum_m = xs_mult16_shl_sat(sum_m, XS_TAB_INVINT(ui - li));
      sum_e = ((frame_exp + (Q::HQ ? 10 : 11)) - shift) - (pre << 1);
    }
    for (int k = li; k < ui; k++) {
      w->nrg_est[2 * (k - base)] = sum_m;
      w->nrg_est[2 * (k - base) + 1] = (int16_t)sum_e;
    }
  }
  cx.sync();
  const int count = j0 < nsf ? cx.uni(tbl[nsf]) - base : 0;
  XS_LANES(c, 0, count) est.own(c) = xs_me(w->nrg_est[2 * c], w->nrg_est[2 * c + 1]);
  cx.sync();
}

/* env_calc.c:1382 */
FX_HD void xs_subbandgain(int16_t e_orig_m, int16_t noise_m, int16_t est_m, int16_t est_e, int16_t noise_e,
                          int16_t ref_e, int sine_present, int sine_mapped, int noise_absc, int16_t *gain,
                          int16_t *noise_floor, int16_t *sine) {
  int16_t v1m, v1e, v2m, v2e, v3m, v3e;
  if (est_m == 0) {
    est_m = 0x4000;
    est_e = 1;
  }
  v1m = xs_mult16_shl_sat(e_orig_m, noise_m);
  v1e = (int16_t)(ref_e + noise_e);
  {
    int32_t accu, d = noise_e - 1;
    if (d >= 0) {
      accu = noise_m + fx_shr(0x4000, d);
      v2e = noise_e;
    } else {
      accu = fx_shr((int32_t)noise_m, -d) + 0x4000;
      v2e = 1;
    }
    if ((accu < 0 ? -accu : accu) >= 0x8000) {
      accu >>= 1;
      v2e++;
    }
    v2m = (int16_t)accu;
  }
  int t = xs_fix_mant_div(v1m, v2m, noise_floor);
  noise_floor[1] = (int16_t)(t + (v1e - v2e) + 1);
  if (sine_present || !noise_absc) {
    v3m = xs_mult16_shl_sat(v2m, est_m);
    v3e = (int16_t)(v2e + est_e);
  } else {
    v3m = est_m;
    v3e = est_e;
  }
  if (!sine_present) {
    v1m = e_orig_m;
    v1e = ref_e;
  }
  t = xs_fix_mant_div(v1m, v3m, gain);
  gain[1] = (int16_t)(t + (v1e - v3e) + 1);
  if (sine_present && sine_mapped) {
    t = xs_fix_mant_div(e_orig_m, v2m, sine);
    sine[1] = (int16_t)(t + (ref_e - v2e) + 1);
  }
}

/* env_calc.c:616 in two steps.  Step 1: for every QMF band covered by the envelope's frequency table
   find its scale-factor band j, its noise-floor band nb and whether its sfb carries a sine in this
   envelope, plus the alias-reduction flag.  The reference finds them walking the tables band by band;
   with strictly increasing tables that start at the same band (ixheaacd_freq_sca.c builds them so)
   the walk has a closed form per band -- j / nb = number of table borders at or below the band, the
   sine flag = OR over the lanes of the same sfb, taken from two ballots -- so lane i does band
   tbl[0] + i.  Step 2 (one band per lane): the three pseudo-float divisions of xs_subbandgain.
   Returns the number of bands at or above max_qmf_subband_aac (elements of v.meta). */
FX_HD int xs_subband_gain_meta(const XsCx &cx, const xaac_sbr_header *h, int max_qmf_subband_aac, const int16_t *tbl,
                               int nsf, int env, XsEnv &v) {
  XsLv tblv, noisev, jn, flags, ar, mt;
  tblv.fill(0);
  noisev.fill(0);
  XS_LANES(i, 0, nsf + 1) tblv.own(i) = tbl[i];
  XS_LANES(i, 0, XAAC_SBR_MAX_NOISE_COEFFS + 1) noisev.own(i) = h->freq_band_tbl_noise[i];
  const int t0 = tblv.get(0);
  int n = tblv.get(nsf) - t0;
  if (n > 64) n = 64;
  const int nnf = cx.uni(h->num_nf_bands);
  jn.fill(0);
  flags.fill(0);
  XS_LANES(i, 0, n) {
    const int k = t0 + i;
    int j = 0, nb = 0;
    XS_UNROLL4
    for (int t = 1; t < nsf; t++) j += tblv.get(t) <= k;
    for (int t = 1; t < nnf; t++) nb += noisev.get(t) <= k;
    jn.own(i) = j | (nb << 8);
  }
  const XsLv jprev = jn.shifted(cx, -1);
  XS_LANES(i, 0, n)
    flags.own(i) = ((i == 0 || (jn.own(i) & 255) != (jprev.own(i) & 255)) ? 1 : 0) |
                   ((env >= v.sine_mapped.own(i)) ? 2 : 0);
  const uint64_t sfb_start = xs_ballot(cx, flags, 1, n), sine_here = xs_ballot(cx, flags, 2, n);
  ar.fill(0);
  mt.fill(0);
  XS_LANES(i, 0, n) {
    const int first = 63 - xs_clz64(sfb_start & xs_mask_upto(i)); /* bit 0 is always set */
    const uint64_t above = sfb_start & ~xs_mask_upto(i);
    const int end = above ? xs_ctz64(above) : n;
    const uint64_t seg = (((uint64_t)1 << end) - 1) & ~(((uint64_t)1 << first) - 1);
    const int present = (sine_here & seg) != 0;
    ar.own(i) = !present;
    mt.own(i) = jn.own(i) | (present << 16);
  }
  /* alias_red is indexed from sub_band_start, meta from max_qmf_subband_aac */
  const int aoff = t0 - cx.uni(h->sub_band_start);
  const XsLv ar_s = ar.shifted(cx, -aoff);
  XS_LANES(e, aoff, aoff + n > 64 ? 64 : aoff + n) v.alias_red.own(e) = ar_s.own(e);
  const int moff = max_qmf_subband_aac > t0 ? max_qmf_subband_aac - t0 : 0;
  const XsLv mt_s = mt.shifted(cx, moff);
  const int n_meta = n > moff ? n - moff : 0;
  XS_LANES(c, 0, n_meta) v.meta.own(c) = mt_s.own(c);
  return n_meta;
}
/* one of the frame's per-envelope lane vectors, by a run-time envelope number */
FX_HD XsLv xs_pick_env(const XsLv *v, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  XsLv r = v[0];
  XS_UNROLL
  for (int k = 1; k < XAAC_SBR_MAX_ENVELOPES; k++) {
    int32_t t = v[k].v;
    asm volatile("" : "+v"(t)); /* keeps the chain a chain of selects: folded into v[i] it would put the array in scratch */
    r.v = i == k ? t : r.v;
  }
  return r;
#else
  return v[i];
#endif
}
/* env_sf: the envelope's scale factors, element j = scale-factor band j */
FX_HD void xs_calc_subband_gains(const XsCx &cx, const XsLv &env_sf, const int16_t *noise_floor, int env, int n_meta,
                                 int skip, XsEnv &v, int noise_absc) {
  const XsLv sm1 = v.sine_mapped.shifted(cx, skip);
  XsLv jv;
  jv.fill(0);
  XS_LANES(c, 0, n_meta) jv.own(c) = v.meta.own(c) & 255;
  const XsLv sfg = env_sf.gather(jv);
  XS_LANES(c, 0, n_meta) {
    const int meta = v.meta.own(c);
    const int16_t sf = (int16_t)sfg.own(c);
    const int16_t nfl = noise_floor[(meta >> 8) & 255];
    const int present = (meta >> 16) & 1;
    const int16_t ref_e = (int16_t)((sf & 63) - 16), ref_m = (int16_t)(sf & 0xffc0);
    const int16_t nm = (int16_t)(nfl & 0xffc0), ne = (int16_t)((nfl & 63) - 38);
    int16_t g[2] = {0, 0}, nl[2] = {0, 0}, sn[2] = {0, 0};
    const int32_t est = v.est.own(c);
    xs_subbandgain(ref_m, nm, xs_m(est), xs_e(est), ne, ref_e, present, env >= sm1.own(c), noise_absc, g, nl, sn);
    v.e_orig.own(c) = xs_me(ref_m, ref_e);
    v.gain.own(c) = xs_me(g[0], g[1]);
    v.noise.own(c) = xs_me(nl[0], nl[1]);
    v.sine.own(c) = xs_me(sn[0], sn[1]);
  }
}

/* env_calc.c:1454: pseudo-float sums over bands [b0,b1) -- order dependent, hence sequential.
   ab[k] = {(m,e) of the first operand, (m,e) of the second}, packed as xs_me. */
FX_HD void xs_avggain(const int32_t (*ab)[2], int b0, int b1, int16_t *o_mant, int16_t *o_exp, int16_t *avg_m,
                      int16_t *avg_e, int flag) {
  int32_t som = 0, soe = 0, sem = 0, see = 0;
  for (int k = b0; k < b1; k++) {
    const int32_t va = ab[k][0], vb = ab[k][1];
    int16_t m = xs_m(va), e = xs_e(va), m2 = xs_m(vb), e2 = xs_e(vb);
    xs_acc_me(&som, &soe, m, e);
    if (flag) {
      m = (int16_t)(((int32_t)m * m2) >> 16);
      e = (int16_t)(e + e2 + 1);
    } else {
      m = m2;
      e = e2;
    }
    xs_acc_me(&sem, &see, m, e);
  }
  int nv = 16 - xs_pnorm32(som);
  if (nv > 0) {
    som >>= nv;
    soe += nv;
  }
  nv = 16 - xs_pnorm32(sem);
  if (nv > 0) {
    sem >>= nv;
    see += nv;
  }
  int16_t so_m, so_e, se_m, se_e;
  if (!flag) {
    so_m = (int16_t)som;
    so_e = (int16_t)soe;
    se_m = (int16_t)sem;
    se_e = (int16_t)see;
  } else {
    se_m = (int16_t)som;
    se_e = (int16_t)soe;
    so_m = (int16_t)sem;
    so_e = (int16_t)see;
  }
  int t = xs_fix_mant_div(so_m, se_m, avg_m);
  *avg_e = (int16_t)(t + (so_e - se_e) + 1);
  *o_mant = so_m;
  *o_exp = so_e;
}

/* env_calc.c:229.  Two kinds of steps alternate: per-band ones (lane = band) and the order-dependent
   pseudo-float sums over a limiter band (lane = limiter band, all limiter bands at once); they hand
   their operands / results over through w->fold_* / w->res_*. */
/* which limiter band each band k (band max_qmf_subband_aac + k) belongs to: tbl_lim[c] <= k + skip < tbl_lim[c + 1]
   (the last such c, as the reference's loop over c would apply them in order; the bands are disjoint and in band
   order); -1 outside.  The same for every envelope of a frame. */
FX_HD XsLv xs_limiter_band_of(const XsCx &cx, const xaac_sbr_header *h, int skip) {
  const int nlf = cx.uni(h->num_lf_bands);
  XsLv limv;
  limv.fill(0);
  XS_LANES(i, 0, nlf + 1) limv.own(i) = h->freq_band_tbl_lim[i];
  XsLv mine;
  mine.fill(-1);
  XS_LANES(k, 0, 64) {
    int c_of = -1;
    for (int c = 0; c < nlf; c++) {
      const int t_lo = limv.get(c), t_hi = limv.get(c + 1);
      const int b0 = t_lo > skip ? t_lo - skip : 0, b1 = t_hi > skip ? t_hi - skip : 0;
      if (k >= b0 && k < b1) c_of = c;
    }
    mine.own(k) = c_of;
  }
  return mine;
}

FX_HD void xs_noiselimiting(const XsCx &cx, const xaac_sbr_header *h, int skip, int n_bands, XsEnv &v, XsWork *w,
                            const int16_t *lim_tab, int noise_absc, const XsLv &band_of) {
  const int16_t lim_m = lim_tab[0], lim_e = lim_tab[1];
  const int nlf = cx.uni(h->num_lf_bands);
  XsLv mine;
  mine.fill(-1);
  XS_LANES(k, 0, n_bands) mine.own(k) = band_of.own(k);
  /* env_calc.c:1454 (avggain, flag 0) per limiter band: sum of e_orig and sum of est -- addends and exponent steps of
     both sums per band here, the two-operation recursions per limiter band below */
  {
    XsLv eo, ee;
    eo.fill(0);
    ee.fill(0);
    XS_LANES(k, 0, n_bands) {
      eo.own(k) = xs_e(v.e_orig.own(k));
      ee.own(k) = xs_e(v.est.own(k));
    }
    const XsLv po = xs_seg_running_max(cx, mine, eo, n_bands), pe = xs_seg_running_max(cx, mine, ee, n_bands);
    XS_LANES(k, 0, n_bands) {
      const int Eo = eo.own(k) > po.own(k) ? eo.own(k) : po.own(k), Ee = ee.own(k) > pe.own(k) ? ee.own(k) : pe.own(k);
      w->fold_b[k][0] = fx_shr(xs_m(v.e_orig.own(k)), Eo - eo.own(k));
      w->fold_b[k][1] = fx_shr(xs_m(v.est.own(k)), Ee - ee.own(k));
      w->fold_b[k][2] = ((Eo - po.own(k)) & 0xff) | (((Ee - pe.own(k)) & 0xff) << 8); /* shr32 masks its count so */
      w->fold_b[k][3] = (Eo & 0xffff) | (int32_t)((uint32_t)Ee << 16);
    }
  }
  cx.sync();
  XS_PAR(c, 0, nlf) {
    const int t_lo = h->freq_band_tbl_lim[c], t_hi = h->freq_band_tbl_lim[c + 1];
    const int b0 = t_lo > skip ? t_lo - skip : 0, b1 = t_hi > skip ? t_hi - skip : 0;
    if (b0 >= b1) continue;
    int32_t som = 0, sem = 0, last = 0;
    XS_UNROLL4
    for (int k = b0; k < b1; k++) {
      const int32_t d = w->fold_b[k][2];
      som = fx_shr(som, d & 255) + w->fold_b[k][0];
      sem = fx_shr(sem, d >> 8) + w->fold_b[k][1];
      last = w->fold_b[k][3];
    }
    int32_t soe = (int16_t)last, see = last >> 16;
    int nv = 16 - xs_pnorm32(som);
    if (nv > 0) {
      som >>= nv;
      soe += nv;
    }
    nv = 16 - xs_pnorm32(sem);
    if (nv > 0) {
      sem >>= nv;
      see += nv;
    }
    const int16_t so_m = (int16_t)som, so_e = (int16_t)soe;
    int16_t mg_m;
    int16_t mg_e = (int16_t)(xs_fix_mant_div(so_m, (int16_t)sem, &mg_m) + (so_e - (int16_t)see) + 1);
    int32_t mt = xs_mult16x16_shl(mg_m, lim_m);
    mg_e = (int16_t)(mg_e + lim_e);
    int tv = fx_norm32(mt);
    mg_e = (int16_t)(mg_e - tv);
    mg_m = (int16_t)(xs_shl(mt, tv) >> 16);
    if (mg_e >= 34) {
      mg_m = 0x3000;
      mg_e = 34;
    }
    w->res_a[c][0] = mg_m;
    w->res_a[c][1] = mg_e;
    w->res_a[c][2] = so_m;
    w->res_a[c][3] = so_e;
  }
  cx.sync();
  XS_T(16);
  /* the gain limit per band; then the sum of what the limited gains, sines and noise deliver (env_calc.c:330-390):
     up to two addends per band, in band order */
  {
    XsLv a_m, a_e, b_me, emax, with_b;
    a_m.fill(0);
    a_e.fill(0);
    b_me.fill(0);
    emax.fill(0);
    with_b.fill(0);
    XS_LANES(k, 0, n_bands) {
      const int c_of = mine.own(k);
      if (c_of < 0) continue;
      const int16_t mg_m = w->res_a[c_of][0], mg_e = w->res_a[c_of][1];
      int16_t gm = xs_m(v.gain.own(k)), ge = xs_e(v.gain.own(k));
      if (ge > mg_e || (ge == mg_e && gm > mg_m)) {
        int16_t na_m;
        int na_e = xs_fix_mant_div(mg_m, gm, &na_m);
        na_e += (mg_e - ge) + 1;
        const int32_t nl = v.noise.own(k);
        v.noise.own(k) =
            xs_me((int16_t)(fx_shl_dir_sat_limit(xs_mult16x16_shl(xs_m(nl), na_m), (int16_t)na_e) >> 16), xs_e(nl));
        gm = mg_m;
        ge = mg_e;
        v.gain.own(k) = xs_me(gm, ge);
      }
      a_m.own(k) = ((int32_t)gm * xs_m(v.est.own(k))) >> 15;
      a_e.own(k) = ge + xs_e(v.est.own(k));
      /* the second addend: the sine if there is one, else the noise unless it is absent (noise_absc) */
      const int32_t sn = v.sine.own(k);
      b_me.own(k) = xs_m(sn) != 0 ? sn : (noise_absc == 0 ? v.noise.own(k) : 0);
      const bool has_b = xs_m(sn) != 0 || noise_absc == 0;
      emax.own(k) = (has_b && xs_e(b_me.own(k)) > a_e.own(k)) ? xs_e(b_me.own(k)) : a_e.own(k);
      with_b.own(k) = has_b ? 1 : 0;
    }
    const XsLv pm = xs_seg_running_max(cx, mine, emax, n_bands);
    XS_LANES(k, 0, n_bands) {
      if (mine.own(k) < 0) continue;
      const int has_b = with_b.own(k);
      const int E0 = pm.own(k);
      const int E1 = a_e.own(k) > E0 ? a_e.own(k) : E0;
      const int eb = xs_e(b_me.own(k));
      const int E2 = (has_b && eb > E1) ? eb : E1;
      w->fold_b[k][0] = fx_shr(a_m.own(k), E1 - a_e.own(k));
      w->fold_b[k][1] = has_b ? fx_shr(xs_m(b_me.own(k)), E2 - eb) : 0;
      w->fold_b[k][2] = ((E1 - E0) & 0xff) | (((E2 - E1) & 0xff) << 8);
      w->fold_b[k][3] = E2;
    }
  }
  cx.sync();
  XS_T(17);
  XS_PAR(c, 0, nlf) {
    const int t_lo = h->freq_band_tbl_lim[c], t_hi = h->freq_band_tbl_lim[c + 1];
    const int b0 = t_lo > skip ? t_lo - skip : 0, b1 = t_hi > skip ? t_hi - skip : 0;
    if (b0 >= b1) continue;
    const int16_t so_m = w->res_a[c][2], so_e = w->res_a[c][3];
    int32_t am = 0, ae = 0;
    XS_UNROLL4
    for (int k = b0; k < b1; k++) {
      const int32_t d = w->fold_b[k][2];
      am = fx_shr(am, d & 255) + w->fold_b[k][0];
      am = fx_shr(am, d >> 8) + w->fold_b[k][1];
      ae = w->fold_b[k][3];
    }
    int nv = 16 - fx_norm32(am);
    if (nv > 0) {
      am >>= nv;
      ae += nv;
    }
    int16_t bg_m;
    int bg_e = xs_fix_mant_div(so_m, (int16_t)am, &bg_m);
    bg_e = (int16_t)(bg_e + (so_e - (int16_t)ae) + 1);
    if (bg_e > 2 || (bg_e == 2 && bg_m > 0x5061)) {
      bg_m = 0x5061;
      bg_e = 2;
    }
    w->res_b[c][0] = bg_m;
    w->res_b[c][1] = (int16_t)bg_e;
  }
  cx.sync();
  XS_T(18);
  XS_LANES(k, 0, n_bands) {
    const int c_of = mine.own(k);
    if (c_of < 0) continue;
    const int16_t bg_m = w->res_b[c_of][0], bg_e = w->res_b[c_of][1];
    const int32_t g = v.gain.own(k), sn = v.sine.own(k), nl = v.noise.own(k);
    v.gain.own(k) = xs_me(xs_mult16_shl(xs_m(g), bg_m), (int16_t)(xs_e(g) + bg_e));
    v.sine.own(k) = xs_me(xs_mult16_shl(xs_m(sn), bg_m), (int16_t)(xs_e(sn) + bg_e));
    v.noise.own(k) = xs_me(xs_mult16_shl(xs_m(nl), bg_m), (int16_t)(xs_e(nl) + bg_e));
  }
  cx.sync();
  XS_T(20);
}

/* env_calc.c:78 (low-power only), step 1: the groups of aliasing bands.  The reference walks the
   bands with a little state machine: a run of bands with (deg[k+1] != 0 && alias_red[k]) is cut into
   groups of four; a run that stops early closes its last group at the stopping band (inclusive when
   that band still has alias_red set), a run that reaches the last band closes it at nsb.  As a
   function of the two bit masks involved that is: a band starts a group iff its offset in its run is
   a multiple of four; its group ends four bands up when the next three bands are still in the run,
   else where the run stops.  Returns the mask of group-start bands; end.own(s) = end of the group
   that starts at s. */
FX_HD uint64_t xs_alias_groups(const XsCx &cx, const XsLv &deg1, const XsLv &alias_red, int nsb, XsLv &end) {
  XsLv f;
  f.fill(0);
  XS_LANES(k, 0, 64)
    f.own(k) = ((k < nsb - 1 && deg1.own(k) != 0 && alias_red.own(k)) ? 1 : 0) | (alias_red.own(k) ? 2 : 0);
  const uint64_t in_run = xs_ballot(cx, f, 1, 64), red = xs_ballot(cx, f, 2, 64);
  XsLv st;
  st.fill(0);
  XS_LANES(k, 0, 64) {
    if (!((in_run >> k) & 1)) continue;
    const uint64_t gaps_below = ~in_run & (xs_mask_upto(k) >> 1);
    const int run_start = gaps_below ? 64 - xs_clz64(gaps_below) : 0;
    if ((k - run_start) & 3) continue;
    st.own(k) = 1;
    int e;
    if (((in_run >> k) & 15) == 15) {
      e = k + 4;
    } else {
      const int z = k + xs_ctz64(~(in_run >> k)); /* first band at or above k outside the run */
      if (z >= nsb - 1)
        e = nsb;
      else
        e = ((red >> z) & 1) ? z + 1 : z;
    }
    end.own(k) = e;
  }
  return xs_ballot(cx, st, 1, 64);
}
/* step 2: equalise the gains inside each group (the lane of a group's first band does its sums) */
FX_HD void xs_alias_reduction(const XsCx &cx, XsEnv &v, XsWork *w, uint64_t starts, const XsLv &end, int nsb) {
  if (starts == 0) return;
  XS_LANES(k, 0, nsb) {
    w->fold_a[k][0] = v.est.own(k);
    w->fold_a[k][1] = v.gain.own(k);
  }
  cx.sync();
  XS_LANES(s, 0, nsb) {
    if (!((starts >> s) & 1)) continue;
    int16_t amp_m, amp_e, gg_m, gg_e;
    xs_avggain(w->fold_a, s, end.own(s), &amp_m, &amp_e, &gg_m, &gg_e, 1);
    w->res_a[s][0] = amp_m;
    w->res_a[s][1] = amp_e;
    w->res_a[s][2] = gg_m;
    w->res_a[s][3] = gg_e;
    w->res_b[s][0] = (int16_t)end.own(s);
  }
  cx.sync();
  XsLv mine;
  mine.fill(-1);
  XS_LANES(k, 0, nsb) {
    const uint64_t below = starts & xs_mask_upto(k);
    if (!below) continue;
    const int s = 63 - xs_clz64(below);
    if (k >= w->res_b[s][0]) continue;
    mine.own(k) = s;
    const int16_t gg_m = w->res_a[s][2], gg_e = w->res_a[s][3];
    int16_t alpha = (int16_t)v.deg.own(k);
    if (k < nsb - 1 && (int16_t)v.deg1.own(k) > alpha) alpha = (int16_t)v.deg1.own(k);
    int32_t gain_m = (int32_t)alpha * gg_m;
    int16_t one_minus = (int16_t)(0x7fff - alpha);
    int32_t tm = xs_m(v.gain.own(k)), te = xs_e(v.gain.own(k));
    tm = ((int32_t)one_minus * tm) >> 15;
    int32_t d = gg_e - te;
    if (d >= 0) {
      te = gg_e;
      tm = fx_shr(tm, d);
      tm = (gain_m >> 15) + tm;
    } else {
      tm = fx_shr(gain_m, 15 - d) + tm;
    }
    v.gain.own(k) = xs_me((int16_t)tm, (int16_t)te);
    /* the reference multiplies the untruncated 32-bit tmp_gain_mant here (env_calc.c:182) */
    w->fold_b[k][0] = (int32_t)((uint32_t)tm * (uint32_t)(int32_t)xs_m(v.est.own(k))) >> 16;
    w->fold_b[k][1] = te + xs_e(v.est.own(k)) + 1;
  }
  cx.sync();
  XS_LANES(s, 0, nsb) {
    if (!((starts >> s) & 1)) continue;
    int32_t mod_m = 0, mod_e = 0;
    const int e = end.own(s);
    for (int k = s; k < e; k++) xs_acc_me(&mod_m, &mod_e, w->fold_b[k][0], w->fold_b[k][1]);
    int nv = 16 - xs_pnorm32(mod_m);
    if (nv > 0) {
      mod_m >>= nv;
      mod_e += nv;
    }
    int16_t comp_m;
    int comp_e = xs_fix_mant_div(w->res_a[s][0], (int16_t)mod_m, &comp_m);
    comp_e = (int16_t)(comp_e + w->res_a[s][1] - (int16_t)mod_e + 1 + 1);
    w->res_b[s][1] = comp_m;
    w->res_a[s][3] = (int16_t)comp_e;
  }
  cx.sync();
  XS_LANES(k, 0, nsb) {
    const int s = mine.own(k);
    if (s < 0) continue;
    const int16_t comp_m = w->res_b[s][1], comp_e = w->res_a[s][3];
    const int32_t g2 = v.gain.own(k);
    v.gain.own(k) = xs_me((int16_t)(((int32_t)xs_m(g2) * comp_m) >> 16), (int16_t)(xs_e(g2) + comp_e));
  }
  cx.sync();
}

/* ---- two envelopes side by side (HQ mode) ----------------------------------------------------------------------------
 * The gain mathematics of an envelope -- xs_subband_gain_meta, xs_calc_subband_gains, xs_noiselimiting,
 * xs_erg_to_amplitude_hq -- only reads the envelope's energy estimates and side info: nothing of it depends on the envelope
 * before.  A frame's SBR range is at most 32 bands wide in every stream the reference's tables produce at 48 kHz and below
 * (26 - 27 bands in the bench's streams), so the chain above leaves more than half of the wave idle and runs once per
 * envelope, 2.9 times per frame in the bench's transient-heavy material.  Here element c of envelope q (q = 0, 1) of a pass
 * lives on lane 32 q + c and the chain runs once per PAIR of envelopes; what differs between the two (frequency resolution,
 * noise-floor row, scale factors, the "no noise" flag of a transient envelope, the envelope number the sine flags are held
 * against, the noise exponent) is a per-lane select.  Same operations in the same order per element -- the sums over a
 * limiter band are the same recursions, on two lane groups.  The slots of the two envelopes are then adjusted one after the
 * other (xs_adapt_noise_gain_hq: the filter buffers chain the envelopes).
 * Taken when the frame is the regular case (xs_pack_frame_ok); every other frame runs the one-envelope chain. */
#define XS_PK 32
struct XsPass {       /* the (wave-uniform) per-envelope values of a pass */
  int n;              /* envelopes in the pass: 1 or 2 */
  int env[2], fr[2], nf_off[2], noise_absc[2], noise_e[2];
};
FX_HD int xs_qsel(int q, const int *v) { return q ? v[1] : v[0]; }

/* xs_energy_per_subband (env_calc.c:1211, complex matrix) for a pass: element (q, c) = band b0 + c over envelope q's slots
   [s0[q], s1[q]).  Both envelopes walk their slots together; a lane whose envelope is the shorter one re-reads its last
   slot (harmless for the maximum) and adds nothing. */
/* One (envelope, band) element of a pass, its slots held in registers: N >= n slots of band k from row `first` on, every
   load in flight before the first use (one LDS latency for the lane's whole column part instead of one per four slots,
   twice), words past the element's own n slots zeroed (they add nothing to the maximum's OR or to the sum).  Returns the
   packed (mantissa, exponent) estimate of env_calc.c:1211.  The OR of the magnitudes has the maximum's leading bit, which is
   all the norm looks at. */
template <int N, class Q>
FX_HD int32_t xs_energy_element_pk(const Q &x, int first, int n, int k, int16_t inv_width, int frame_exp2) {
  int32_t re[N], im[N];
  XS_UNROLL
  for (int j = 0; j < N; j++) {
    const int row = first + (j < n ? j : (n > 0 ? n - 1 : 0)); /* (an element without slots -- borders out of order -- reads its first row: masked below) */
    re[j] = x(row, k);
    if constexpr (Q::HQ) im[j] = x.im(row, k); else im[j] = 0;
  }
  int32_t mx = 1;
  XS_UNROLL
  for (int j = 0; j < N; j++) {
    XS_KEEP(re[j]); /* (the loads stay unconditional and together: as operands of a select alone each was sunk into a
                       predicated region of its own, with a full wait inside) */
    if constexpr (Q::HQ) XS_KEEP(im[j]);
    re[j] = j < n ? re[j] : 0;
    im[j] = j < n ? im[j] : 0;
    mx |= fx_abs_nrm(re[j]);
    if constexpr (Q::HQ) mx |= fx_abs_nrm(im[j]);
  }
  const int pre = xs_pnorm32(mx) - (Q::HQ ? 4 : 3);
  int shift = 16 - pre;
  const int e_shr = shift > 0 ? shift & 31 : 0, e_shl = shift > 0 ? 0 : (-shift) & 31;
  int32_t accu = 0;
  XS_UNROLL
  for (int j = 0; j < N; j++) {
    int16_t t = (int16_t)((int32_t)((uint32_t)re[j] << e_shl) >> e_shr);
    accu = fx_add(accu, (int32_t)t * t);
    if constexpr (Q::HQ) {
      t = (int16_t)((int32_t)((uint32_t)im[j] << e_shl) >> e_shr);
      accu = fx_add(accu, (int32_t)t * t);
    }
  }
  if (accu == 0) return 0;
  shift = -xs_pnorm32(accu);
  int16_t sum_m = (int16_t)xs_shr_dir_sat_limit(accu, 16 + shift);
  sum_m = xs_mult16_shl_sat(sum_m, inv_width);
  shift = shift - (pre << 1) + (Q::HQ ? 0 : 1);
  return xs_me(sum_m, (int16_t)(frame_exp2 + shift + 1));
}
/* the same for any number of slots: two walks over the column part, eight slots' loads in flight at a time */
template <class Q>
FX_HD int32_t xs_energy_element_long(const Q &x, int first, int n, int nmax, int k, int16_t inv_width, int frame_exp2) {
  int32_t mx = 1;
  for (int j0 = 0; j0 < nmax; j0 += 8) {
    int32_t re[8], im[8];
    XS_UNROLL
    for (int j = 0; j < 8; j++) {
      const int row = first + (j0 + j < n ? j0 + j : (n > 0 ? n - 1 : 0));
      re[j] = x(row, k);
      if constexpr (Q::HQ) im[j] = x.im(row, k); else im[j] = 0;
    }
    XS_UNROLL
    for (int j = 0; j < 8; j++) {
      XS_KEEP(re[j]);
      if constexpr (Q::HQ) XS_KEEP(im[j]);
      mx |= fx_abs_nrm(re[j]); /* (a slot read twice changes no maximum) */
      if constexpr (Q::HQ) mx |= fx_abs_nrm(im[j]);
    }
  }
  const int pre = xs_pnorm32(mx) - (Q::HQ ? 4 : 3);
  int shift = 16 - pre;
  const int e_shr = shift > 0 ? shift & 31 : 0, e_shl = shift > 0 ? 0 : (-shift) & 31;
  int32_t accu = 0;
  for (int j0 = 0; j0 < nmax; j0 += 8) {
    int32_t re[8], im[8];
    XS_UNROLL
    for (int j = 0; j < 8; j++) {
      const int row = first + (j0 + j < n ? j0 + j : (n > 0 ? n - 1 : 0));
      re[j] = x(row, k);
      if constexpr (Q::HQ) im[j] = x.im(row, k); else im[j] = 0;
    }
    XS_UNROLL
    for (int j = 0; j < 8; j++) {
      XS_KEEP(re[j]);
      if constexpr (Q::HQ) XS_KEEP(im[j]);
      re[j] = j0 + j < n ? re[j] : 0;
      im[j] = j0 + j < n ? im[j] : 0;
      int16_t t = (int16_t)((int32_t)((uint32_t)re[j] << e_shl) >> e_shr);
      accu = fx_add(accu, (int32_t)t * t);
      if constexpr (Q::HQ) {
        t = (int16_t)((int32_t)((uint32_t)im[j] << e_shl) >> e_shr);
        accu = fx_add(accu, (int32_t)t * t);
      }
    }
  }
  if (accu == 0) return 0;
  shift = -xs_pnorm32(accu);
  int16_t sum_m = (int16_t)xs_shr_dir_sat_limit(accu, 16 + shift);
  sum_m = xs_mult16_shl_sat(sum_m, inv_width);
  shift = shift - (pre << 1) + (Q::HQ ? 0 : 1);
  return xs_me(sum_m, (int16_t)(frame_exp2 + shift + 1));
}
template <class Q>
FX_HD void xs_energy_per_subband_pk(const XsCx &cx, const Q &x, const XsPass &ps, const int *s0, const int *s1, int b0,
                                    int nb, int frame_exp, XsLv &est) {
  const int n0 = s1[0] - s0[0], n1 = ps.n > 1 ? s1[1] - s0[1] : 0, nmax = n0 > n1 ? n0 : n1;
  const int frame_exp2 = frame_exp << 1;
  XS_LANES(l, 0, 64) {
    const int q = l / XS_PK, c = l & (XS_PK - 1);
    int32_t e = 0;
    if (q < ps.n && c < nb) {
      const int first = xs_qsel(q, s0), n = q ? n1 : n0, k = b0 + c;
      const int16_t inv_width = XS_TAB_INVINT(n > 0 ? n : 0);
      if (nmax <= 8) /* (uniform) */
        e = xs_energy_element_pk<8>(x, first, n, k, inv_width, frame_exp2);
      else if (nmax <= 16)
        e = xs_energy_element_pk<16>(x, first, n, k, inv_width, frame_exp2);
      else
        e = xs_energy_element_long(x, first, n, nmax, k, inv_width, frame_exp2);
    }
    est.own(l) = e;
  }
}

/* per frame: for each band of the SBR range (element i = band sub_band_start + i) its scale-factor band under the high- and
   the low-resolution table and its noise-floor band (the table walks of xs_subband_gain_meta, once per frame instead of
   once per envelope): j | nb << 8 */
FX_HD void xs_band_maps(const XsCx &cx, const xaac_sbr_header *h, int nsb, XsLv &jn_hi, XsLv &jn_lo) {
  XsLv thi, tlo, noisev;
  thi.fill(0);
  tlo.fill(0);
  noisev.fill(0);
  const int nhi = cx.uni(h->num_sf_bands[1]), nlo = cx.uni(h->num_sf_bands[0]), nnf = cx.uni(h->num_nf_bands);
  XS_LANES(i, 0, nhi + 1) thi.own(i) = h->freq_band_tbl_hi[i];
  XS_LANES(i, 0, nlo + 1) tlo.own(i) = h->freq_band_tbl_lo[i];
  XS_LANES(i, 0, XAAC_SBR_MAX_NOISE_COEFFS + 1) noisev.own(i) = h->freq_band_tbl_noise[i];
  const int t0 = thi.get(0);
  jn_hi.fill(0);
  jn_lo.fill(0);
  XS_LANES(i, 0, nsb) {
    const int k = t0 + i;
    int jh = 0, jl = 0, nb = 0;
    XS_UNROLL4
    for (int t = 1; t < nhi; t++) jh += thi.get(t) <= k;
    for (int t = 1; t < nlo; t++) jl += tlo.get(t) <= k;
    for (int t = 1; t < nnf; t++) nb += noisev.get(t) <= k;
    jn_hi.own(i) = jh | (nb << 8);
    jn_lo.own(i) = jl | (nb << 8);
  }
}

/* xs_subband_gain_meta for a pass.  jn_*: xs_band_maps; fills v.meta (element (q, c): band max_qmf_subband_aac + c of
   envelope q).  skip = max_qmf_subband_aac - sub_band_start, nsb = sub_band_end - sub_band_start <= 32. */
FX_HD void xs_subband_gain_meta_pk(const XsCx &cx, const XsPass &ps, const XsLv &jn_hi, const XsLv &jn_lo, int nsb, int skip,
                                   XsEnv &v) {
  XsLv idx, flags;
  idx.fill(0);
  XS_LANES(l, 0, 64) idx.own(l) = l & (XS_PK - 1);
  const XsLv gh = jn_hi.gather(idx), gl = jn_lo.gather(idx), sm = v.sine_mapped.gather(idx);
  XsLv jn;
  jn.fill(0);
  flags.fill(0);
  XS_LANES(l, 0, 64) {
    const int q = l / XS_PK, i = l & (XS_PK - 1);
    if (q < ps.n && i < nsb) jn.own(l) = xs_qsel(q, ps.fr) ? gh.own(l) : gl.own(l);
  }
  const XsLv jprev = jn.shifted(cx, -1);
  XS_LANES(l, 0, 64) {
    const int q = l / XS_PK, i = l & (XS_PK - 1);
    if (q < ps.n && i < nsb)
      flags.own(l) = ((i == 0 || (jn.own(l) & 255) != (jprev.own(l) & 255)) ? 1 : 0) | ((xs_qsel(q, ps.env) >= sm.own(l)) ? 2 : 0) | 4;
  }
  const uint64_t sfb_start = xs_ballot(cx, flags, 1, 64), sine_here = xs_ballot(cx, flags, 2, 64);
  XsLv mt, ar;
  mt.fill(0);
  ar.fill(0);
  XS_LANES(l, 0, 64) {
    const int q = l / XS_PK, i = l & (XS_PK - 1);
    if (q < ps.n && i < nsb) {
      const uint64_t st_q = (sfb_start >> (XS_PK * q)) & 0xffffffffull, sn_q = (sine_here >> (XS_PK * q)) & 0xffffffffull;
      const int first = 63 - xs_clz64(st_q & xs_mask_upto(i)); /* bit 0 is always set */
      const uint64_t above = st_q & ~xs_mask_upto(i);
      const int end = above ? xs_ctz64(above) : nsb;
      const uint64_t seg = (((uint64_t)1 << end) - 1) & ~(((uint64_t)1 << first) - 1);
      const int present = (sn_q & seg) != 0;
      mt.own(l) = jn.own(l) | (present << 16);
      ar.own(l) = !present;
    }
  }
  v.alias_red = ar; /* (low-power passes: element (q, i) = band sub_band_start + i of envelope q; nothing reads it in HQ mode) */
  /* meta is indexed from max_qmf_subband_aac: element (q, c) = band (q, c + skip) */
  XsLv src;
  src.fill(0);
  XS_LANES(l, 0, 64) src.own(l) = (l & (XS_PK - 1)) + skip < XS_PK ? l + skip : l;
  const XsLv mt_s = mt.gather(src);
  XS_LANES(l, 0, 64) {
    const int c = l & (XS_PK - 1);
    v.meta.own(l) = (c + skip < nsb) ? mt_s.own(l) : 0;
  }
}

/* xs_calc_subband_gains for a pass: sf_a / sf_b = the two envelopes' scale factors (element j = scale-factor band j) */
FX_HD void xs_calc_subband_gains_pk(const XsCx &cx, const XsPass &ps, const XsLv &sf_a, const XsLv &sf_b,
                                    const int16_t *noise_floor_all, int bands, int skip, XsEnv &v) {
  XsLv idx, jv;
  idx.fill(0);
  jv.fill(0);
  XS_LANES(l, 0, 64) {
    idx.own(l) = ((l & (XS_PK - 1)) + skip) & 63;
    jv.own(l) = v.meta.own(l) & 255;
  }
  const XsLv sm1 = v.sine_mapped.gather(idx); /* element (q, c) = sine_mapped[c + skip] */
  const XsLv ga = sf_a.gather(jv), gb = sf_b.gather(jv);
  XS_LANES(l, 0, 64) {
    const int q = l / XS_PK, c = l & (XS_PK - 1);
    if (q < ps.n && c < bands) {
      const int meta = v.meta.own(l);
      const int16_t sf = (int16_t)(q ? gb.own(l) : ga.own(l));
      const int16_t nfl = noise_floor_all[xs_qsel(q, ps.nf_off) + ((meta >> 8) & 255)];
      const int present = (meta >> 16) & 1;
      const int16_t ref_e = (int16_t)((sf & 63) - 16), ref_m = (int16_t)(sf & 0xffc0);
      const int16_t nm = (int16_t)(nfl & 0xffc0), ne = (int16_t)((nfl & 63) - 38);
      int16_t g[2] = {0, 0}, nl[2] = {0, 0}, sn[2] = {0, 0};
      const int32_t est = v.est.own(l);
      const int mapped = c + skip < 64 ? (xs_qsel(q, ps.env) >= sm1.own(l)) : (xs_qsel(q, ps.env) >= 0);
      xs_subbandgain(ref_m, nm, xs_m(est), xs_e(est), ne, ref_e, present, mapped, xs_qsel(q, ps.noise_absc), g, nl, sn);
      v.e_orig.own(l) = xs_me(ref_m, ref_e);
      v.gain.own(l) = xs_me(g[0], g[1]);
      v.noise.own(l) = xs_me(nl[0], nl[1]);
      v.sine.own(l) = xs_me(sn[0], sn[1]);
    }
  }
}

/* xs_noiselimiting for a pass.  band_of: xs_limiter_band_of (element c = band max_qmf_subband_aac + c).  The limiter bands
   of envelope q are the recursion lanes 16 q .. 16 q + nlf - 1 and use rows 16 q + c of res_a / res_b; a band's operands sit
   in row 32 q + c of fold_b. */
FX_HD void xs_noiselimiting_pk(const XsCx &cx, const xaac_sbr_header *h, const XsPass &ps, int skip, int bands, XsEnv &v,
                               XsWork *w, const int16_t *lim_tab, const XsLv &band_of) {
  const int16_t lim_m = lim_tab[0], lim_e = lim_tab[1];
  const int nlf = cx.uni(h->num_lf_bands);
  XsLv idx, mine, key;
  idx.fill(0);
  XS_LANES(l, 0, 64) idx.own(l) = l & (XS_PK - 1);
  const XsLv bo = band_of.gather(idx);
  mine.fill(-1);
  key.fill(-1);
  XS_LANES(l, 0, 64) {
    const int q = l / XS_PK, c = l & (XS_PK - 1);
    if (q < ps.n && c < bands) {
      mine.own(l) = bo.own(l);
      key.own(l) = bo.own(l) < 0 ? -1 : 16 * q + bo.own(l);
    }
  }
  {
    XsLv eo, ee;
    eo.fill(0);
    ee.fill(0);
    XS_LANES(l, 0, 64) {
      eo.own(l) = xs_e(v.e_orig.own(l));
      ee.own(l) = xs_e(v.est.own(l));
    }
    const XsLv po = xs_seg_running_max(cx, key, eo, 64), pe = xs_seg_running_max(cx, key, ee, 64);
    XS_LANES(l, 0, 64) {
      const int Eo = eo.own(l) > po.own(l) ? eo.own(l) : po.own(l), Ee = ee.own(l) > pe.own(l) ? ee.own(l) : pe.own(l);
      w->fold_b[l][0] = fx_shr(xs_m(v.e_orig.own(l)), Eo - eo.own(l));
      w->fold_b[l][1] = fx_shr(xs_m(v.est.own(l)), Ee - ee.own(l));
      w->fold_b[l][2] = ((Eo - po.own(l)) & 0xff) | (((Ee - pe.own(l)) & 0xff) << 8);
      w->fold_b[l][3] = (Eo & 0xffff) | (int32_t)((uint32_t)Ee << 16);
    }
  }
  cx.sync();
  XS_T(16);
  XS_LANES(r, 0, 32) {
    const int q = r >> 4, c = r & 15;
    if (q < ps.n && c < nlf) {
      const int t_lo = h->freq_band_tbl_lim[c], t_hi = h->freq_band_tbl_lim[c + 1];
      int b0 = t_lo > skip ? t_lo - skip : 0, b1 = t_hi > skip ? t_hi - skip : 0;
      if (b1 > bands) b1 = bands; /* (the bands of the pass: xs_pack_frame_ok has checked that the table ends there) */
      if (b0 < b1) {
        int32_t som = 0, sem = 0, last = 0;
        XS_UNROLL4
        for (int k = b0; k < b1; k++) {
          const int32_t d = w->fold_b[XS_PK * q + k][2];
          som = fx_shr(som, d & 255) + w->fold_b[XS_PK * q + k][0];
          sem = fx_shr(sem, d >> 8) + w->fold_b[XS_PK * q + k][1];
          last = w->fold_b[XS_PK * q + k][3];
        }
        int32_t soe = (int16_t)last, see = last >> 16;
        int nv = 16 - xs_pnorm32(som);
        if (nv > 0) {
          som >>= nv;
          soe += nv;
        }
        nv = 16 - xs_pnorm32(sem);
        if (nv > 0) {
          sem >>= nv;
          see += nv;
        }
        const int16_t so_m = (int16_t)som, so_e = (int16_t)soe;
        int16_t mg_m;
        int16_t mg_e = (int16_t)(xs_fix_mant_div(so_m, (int16_t)sem, &mg_m) + (so_e - (int16_t)see) + 1);
        int32_t mt = xs_mult16x16_shl(mg_m, lim_m);
        mg_e = (int16_t)(mg_e + lim_e);
        int tv = fx_norm32(mt);
        mg_e = (int16_t)(mg_e - tv);
        mg_m = (int16_t)(xs_shl(mt, tv) >> 16);
        if (mg_e >= 34) {
          mg_m = 0x3000;
          mg_e = 34;
        }
        w->res_a[r][0] = mg_m;
        w->res_a[r][1] = mg_e;
        w->res_a[r][2] = so_m;
        w->res_a[r][3] = so_e;
      }
    }
  }
  cx.sync();
  XS_T(17);
  {
    XsLv a_m, a_e, b_me, emax, with_b;
    a_m.fill(0);
    a_e.fill(0);
    b_me.fill(0);
    emax.fill(0);
    with_b.fill(0);
    XS_LANES(l, 0, 64) {
      const int c_of = mine.own(l);
      if (c_of < 0) continue;
      const int q = l / XS_PK, noise_absc = xs_qsel(q, ps.noise_absc);
      const int16_t mg_m = w->res_a[16 * q + c_of][0], mg_e = w->res_a[16 * q + c_of][1];
      int16_t gm = xs_m(v.gain.own(l)), ge = xs_e(v.gain.own(l));
      if (ge > mg_e || (ge == mg_e && gm > mg_m)) {
        int16_t na_m;
        int na_e = xs_fix_mant_div(mg_m, gm, &na_m);
        na_e += (mg_e - ge) + 1;
        const int32_t nl = v.noise.own(l);
        v.noise.own(l) =
            xs_me((int16_t)(fx_shl_dir_sat_limit(xs_mult16x16_shl(xs_m(nl), na_m), (int16_t)na_e) >> 16), xs_e(nl));
        gm = mg_m;
        ge = mg_e;
        v.gain.own(l) = xs_me(gm, ge);
      }
      a_m.own(l) = ((int32_t)gm * xs_m(v.est.own(l))) >> 15;
      a_e.own(l) = ge + xs_e(v.est.own(l));
      const int32_t sn = v.sine.own(l);
      b_me.own(l) = xs_m(sn) != 0 ? sn : (noise_absc == 0 ? v.noise.own(l) : 0);
      const bool has_b = xs_m(sn) != 0 || noise_absc == 0;
      emax.own(l) = (has_b && xs_e(b_me.own(l)) > a_e.own(l)) ? xs_e(b_me.own(l)) : a_e.own(l);
      with_b.own(l) = has_b ? 1 : 0;
    }
    const XsLv pm = xs_seg_running_max(cx, key, emax, 64);
    XS_LANES(l, 0, 64) {
      if (mine.own(l) < 0) continue;
      const int has_b = with_b.own(l);
      const int E0 = pm.own(l);
      const int E1 = a_e.own(l) > E0 ? a_e.own(l) : E0;
      const int eb = xs_e(b_me.own(l));
      const int E2 = (has_b && eb > E1) ? eb : E1;
      w->fold_b[l][0] = fx_shr(a_m.own(l), E1 - a_e.own(l));
      w->fold_b[l][1] = has_b ? fx_shr(xs_m(b_me.own(l)), E2 - eb) : 0;
      w->fold_b[l][2] = ((E1 - E0) & 0xff) | (((E2 - E1) & 0xff) << 8);
      w->fold_b[l][3] = E2;
    }
  }
  cx.sync();
  XS_T(18);
  XS_LANES(r, 0, 32) {
    const int q = r >> 4, c = r & 15;
    if (q < ps.n && c < nlf) {
      const int t_lo = h->freq_band_tbl_lim[c], t_hi = h->freq_band_tbl_lim[c + 1];
      int b0 = t_lo > skip ? t_lo - skip : 0, b1 = t_hi > skip ? t_hi - skip : 0;
      if (b1 > bands) b1 = bands;
      if (b0 < b1) {
        const int16_t so_m = w->res_a[r][2], so_e = w->res_a[r][3];
        int32_t am = 0, ae = 0;
        XS_UNROLL4
        for (int k = b0; k < b1; k++) {
          const int32_t d = w->fold_b[XS_PK * q + k][2];
          am = fx_shr(am, d & 255) + w->fold_b[XS_PK * q + k][0];
          am = fx_shr(am, d >> 8) + w->fold_b[XS_PK * q + k][1];
          ae = w->fold_b[XS_PK * q + k][3];
        }
        int nv = 16 - fx_norm32(am);
        if (nv > 0) {
          am >>= nv;
          ae += nv;
        }
        int16_t bg_m;
        int bg_e = xs_fix_mant_div(so_m, (int16_t)am, &bg_m);
        bg_e = (int16_t)(bg_e + (so_e - (int16_t)ae) + 1);
        if (bg_e > 2 || (bg_e == 2 && bg_m > 0x5061)) {
          bg_m = 0x5061;
          bg_e = 2;
        }
        w->res_b[r][0] = bg_m;
        w->res_b[r][1] = (int16_t)bg_e;
      }
    }
  }
  cx.sync();
  XS_T(19);
  XS_LANES(l, 0, 64) {
    const int c_of = mine.own(l);
    if (c_of < 0) continue;
    const int q = l / XS_PK;
    const int16_t bg_m = w->res_b[16 * q + c_of][0], bg_e = w->res_b[16 * q + c_of][1];
    const int32_t g = v.gain.own(l), sn = v.sine.own(l), nl = v.noise.own(l);
    v.gain.own(l) = xs_me(xs_mult16_shl(xs_m(g), bg_m), (int16_t)(xs_e(g) + bg_e));
    v.sine.own(l) = xs_me(xs_mult16_shl(xs_m(sn), bg_m), (int16_t)(xs_e(sn) + bg_e));
    v.noise.own(l) = xs_me(xs_mult16_shl(xs_m(nl), bg_m), (int16_t)(xs_e(nl) + bg_e));
  }
  cx.sync();
  XS_T(20);
}

/* env_calc.c:423 */
FX_HD void xs_erg_to_amplitude_lp(const XsCx &cx, int bands, int16_t noise_e, XsEnv &v) {
  XS_LANES(k, 0, bands) {
    int16_t sn[2] = {xs_m(v.sine.own(k)), xs_e(v.sine.own(k))};
    int16_t g[2] = {xs_m(v.gain.own(k)), xs_e(v.gain.own(k))};
    int16_t nl[2] = {xs_m(v.noise.own(k)), xs_e(v.noise.own(k))};
    xs_mant_exp_sqrt(sn);
    xs_mant_exp_sqrt(g);
    xs_mant_exp_sqrt(nl);
    int shift = (noise_e - nl[1]) - 4;
    if (shift > 0)
      nl[0] = (int16_t)xs_sar(nl[0], shift);
    else
      nl[0] = (int16_t)xs_shl(nl[0], -shift);
    shift = sn[1] - noise_e;
    if (shift > 0)
      sn[0] = xs_shl16_sat(sn[0], (int16_t)shift);
    else
      sn[0] = (int16_t)xs_sar(sn[0], (int16_t)-shift);
    v.sine.own(k) = xs_me(sn[0], sn[1]);
    v.gain.own(k) = xs_me(g[0], g[1]);
    v.noise.own(k) = xs_me(nl[0], nl[1]);
  }
}

/* env_calc.c:1017, one band */
FX_HD void xs_equalize_filt_buf(int16_t *fb, int16_t *gain) {
  int32_t fe = fb[1], ge = gain[1], fm = fb[0], gm = gain[0];
  int32_t diff = ge - fe;
  if (diff >= 0) {
    fb[1] = (int16_t)ge;
    fb[0] = (int16_t)xs_sar(fb[0], diff);
  } else {
    int32_t reserve = fx_norm32(fm) - 16;
    if (diff + reserve >= 0) {
      fb[0] = (int16_t)xs_shl(fm, -diff);
      fb[1] = (int16_t)(fe + diff);
    } else {
      fb[0] = (int16_t)xs_shl(fm, reserve);
      fb[1] = (int16_t)(fe - reserve);
      int32_t shift = -(reserve + diff);
      gain[0] = (int16_t)xs_sar(gm, shift);
      gain[1] = (int16_t)(gain[1] + shift);
    }
  }
}

/* env_calc.c:1080, one value */
FX_HD int16_t xs_noise_rescale(int16_t v, int diff) {
  if (diff > 0) return (int16_t)xs_sar(v, diff);
  if (diff < 0) return (int16_t)xs_shl(v, -diff);
  return v;
}

#define XS_FACTOR ((int32_t)(0x010b0000 * 2))

/* env_calc.c:479 with :1564 (harmonic index 0 / 2) and :1617 (1 / 3), low-power branch: apply gains,
   noise and sines to slots [s0,s1).
   The reference loops slots outside, bands inside.  A band only ever touches its own column of x
   and its own gain / noise / sine / filter-buffer entries -- except that in the odd-index slots band 0
   also adds a sine tail to column b0-1 and the last band to column b0+nsb, which no band owns -- so
   the loops are interchanged: each lane keeps its band's constants in registers and walks the slots.
   The odd-index code of the reference walks the bands carrying along the neighbours' sine levels, a
   sign that alternates from band 1 on (freq_inv * (-1)^(k-1)) and the number of sines seen so far;
   spelled out per band, what is added to the gained sample is a per-envelope constant `term1` (its
   negative when the harmonic index is 3).  Shifts by 0 are the identity in both directions, which
   is why the reference's "> 0" / ">= 0" variants need no distinction. */
template <class ST, class Q>
FX_HD void xs_adapt_noise_gain_lp(const XsCx &cx, ST *st, XsEnv &v, const int16_t *rand_hi, int noise_e, int nsb,
                                  int skip, int s0, int s1, int input_e, int adj_e, int final_e, int sb_start,
                                  int lb_scale, int noise_absc, const Q &x) {
  const int bands = nsb - skip;
  const int start_up = cx.uni(st->start_up);
  const int ph0 = cx.uni(st->ph_index), harm0 = cx.uni(st->harm_index);
  const int fb_noise_e0 = start_up ? noise_e : cx.uni(st->filt_buf_noise_e);
  cx.sync(); /* everyone has read the scalars before they are updated */
  XS_LANES(k, 0, bands) {
    int16_t g[2] = {xs_m(v.gain.own(k)), xs_e(v.gain.own(k))};
    if (start_up) {
      st->filt_buf_me[2 * (skip + k)] = g[0];
      st->filt_buf_me[2 * (skip + k) + 1] = g[1];
      st->filt_buf_noise_m[skip + k] = xs_m(v.noise.own(k));
    } else {
      xs_equalize_filt_buf(&st->filt_buf_me[2 * (skip + k)], g);
      v.gain.own(k) = xs_me(g[0], g[1]);
    }
  }
  cx.sync();
  const XsLv tone = xs_prefix_nonzero_m(cx, v.sine, nsb);
  const XsLv s_prev = v.sine.shifted(cx, -1), s_next = v.sine.shifted(cx, 1);
  XS_T(21);
  const int nm1 = nsb - 1;
  int fi0 = !(sb_start & 1); /* freq_inv for harmonic index 1; index 3 negates it */
  fi0 = (fi0 << 1) - 1;
  XS_LANES(k, 0, nsb) {
    const int16_t gm = xs_m(v.gain.own(k)), ge = xs_e(v.gain.own(k));
    const int16_t sl = xs_m(v.sine.own(k)), sl_prev = xs_m(s_prev.own(k));
    const int16_t sl_next = (k + 1 < nsb) ? xs_m(s_next.own(k)) : (int16_t)0;
    int16_t nl = xs_m(v.noise.own(k));
    const int with_noise = !noise_absc && sl == 0;
    const int few_tones = tone.own(k) <= 16;
    const int32_t sine32 = xs_shl(sl, 16);
    /* odd harmonic index: what band k adds to its own sample when the index is 1 */
    int32_t term1;
    if (k == 0) {
      term1 = fx_mul32x16(XS_FACTOR, sl_next);
      if (fi0 < 0) term1 = -term1;
    } else if (k < nm1) {
      const int32_t add = fx_mul32x16(XS_FACTOR, (int16_t)(sl_prev - sl_next));
      term1 = few_tones ? (((k & 1) ? fi0 : -fi0) < 0 ? -add : add) : 0;
    } else {
      const int32_t tms = fx_mul32x16(XS_FACTOR, sl_prev);
      term1 = few_tones ? (((nm1 & 1) ? fi0 : -fi0) > 0 ? tms : -tms) : 0;
    }
    /* ... and what the two edge bands add to the column outside the range (index 1; sign as above) */
    int32_t edge1 = 0;
    int edge_col = -1;
    if (k == 0) {
      edge1 = fx_mul32x16(XS_FACTOR, sl); /* shifted by the slot's noise exponent below */
      edge_col = sb_start - 1;
    }
    if (k == nm1 && nm1 > 0 && few_tones && k + sb_start < 62) {
      const int32_t tm2 = fx_mul32x16(XS_FACTOR, sl);
      edge1 = ((nm1 & 1) ? fi0 : -fi0) > 0 ? -tm2 : tm2;
      edge_col = sb_start + k + 1;
    }
    int16_t fbn = st->filt_buf_noise_m[k];
    int ne = noise_e, fbe = fb_noise_e0, ph = ph0, harm = harm0;
    /* the gain's shift of a slot, ge - (scale_change - 1) with one of two scale changes, as a (left, right) pair of counts of
       which one is zero: lane constants, so that the slot loop has no branch on the lane's exponent (xs_shl / xs_sar take
       their counts modulo 32) */
    const int sh_a = ge - ((adj_e - input_e) - 1), sh_b = ge - ((final_e - input_e) - 1);
    const int shl_a = sh_a > 0 ? (sh_a & 31) : 0, shr_a = sh_a > 0 ? 0 : ((-sh_a) & 31);
    const int shl_b = sh_b > 0 ? (sh_b & 31) : 0, shr_b = sh_b > 0 ? 0 : ((-sh_b) & 31);
    for (int l = s0; l < s1; l++) {
      if (l == 32 && s0 < 32) { /* (uniform) */
        const int diff = final_e - ne;
        ne = final_e;
        const int16_t nl2 = xs_noise_rescale(nl, diff);
        nl = k < bands ? nl2 : nl;
      }
      fbn = xs_noise_rescale(fbn, fbe - ne);
      fbe = ne;
      const int16_t rp = rand_hi[ph + 1 + k];
      const int hi = harm;
      ph = (ph + nsb) & 511;
      harm = (harm + 1) & 3;
      int32_t val = fx_mul32x16(x(l, sb_start + k), gm);
      val = (int32_t)((uint32_t)val << (l < 32 ? shl_a : shl_b)) >> (l < 32 ? shr_a : shr_b);
      const int32_t noisy = xs_mac16x16_shl_sat(val, rp, nl); /* (every lane: a select below, not a branch per lane) */
      if (!(hi & 1)) {
        const int32_t toned = hi == 0 ? fx_add_sat(val, sine32) : fx_sub_sat(val, sine32);
        val = with_noise ? noisy : toned;
      } else {
        val = with_noise ? noisy : val;
        val = fx_add_sat(val, hi == 1 ? term1 : -term1);
        if (edge_col >= 0) {
          int32_t t = edge1;
          const int neg = (hi == 1 ? fi0 : -fi0) < 0;
          if (k == 0) { /* band 0's tail carries the noise exponent; freq_inv < 0 adds it, else subtracts */
            const int16_t nexp = (int16_t)((ne - 16) - lb_scale);
            t = nexp > 0 ? fx_shl(t, nexp) : fx_shr(t, -nexp);
            x(l, edge_col) = neg ? fx_add_sat(x(l, edge_col), t) : fx_sub_sat(x(l, edge_col), t);
          } else { /* edge1 holds the index-1 sign already */
            x(l, edge_col) = hi == 1 ? fx_add_sat(x(l, edge_col), t) : fx_sub_sat(x(l, edge_col), t);
          }
        }
      }
      x(l, sb_start + k) = val;
    }
    st->filt_buf_noise_m[k] = fbn;
    v.noise.own(k) = xs_me(nl, xs_e(v.noise.own(k)));
  }
  cx.sync();
  XS_T(22);
  XS_LANES(k, 0, bands) {
    st->filt_buf_me[2 * (skip + k)] = xs_m(v.gain.own(k));
    st->filt_buf_noise_m[skip + k] = xs_m(v.noise.own(k));
  }
  XS_ONE {
    const int n = s1 > s0 ? s1 - s0 : 0;
    int ne = noise_e;
    if (s0 < 32 && s1 > 32) ne = final_e;
    st->start_up = 0;
    st->filt_buf_noise_e = n > 0 ? ne : fb_noise_e0;
    st->ph_index = (int16_t)((ph0 + n * nsb) & 511);
    st->harm_index = (int16_t)((harm0 + n) & 3);
  }
  cx.sync();
}

/* xs_adapt_noise_gain_lp for an envelope of a pass (nsb <= 32; the envelope's gains, noise and sine levels on lanes
   off .. off + nsb - 1 of v): the slots of the envelope on the two halves of the wave, lane 32 h + k = band k, the first
   n_half slots on the lower half, the others on the upper one.  Nothing in the low-power slot loop runs from slot to slot -- the phase index and the harmonic
   index are s0's plus a multiple of the slot number, the noise level and the noise exponent change once, where the slots
   pass 32 (env_calc.c:560-575), and the noise filter buffer is only rescaled along the way --, and a slot's edge columns
   are touched by that slot alone, so each half does its share of the slots and the state that is left behind (filter buffers,
   indices) is what the slot-by-slot walk leaves. */
template <class ST, class Q>
FX_HD void xs_adapt_noise_gain_lp_split(const XsCx &cx, ST *st, const XsEnv &v, int off, const int16_t *rand_hi, int noise_e,
                                        int nsb, int skip, int s0, int s1, int input_e, int adj_e, int final_e, int sb_start,
                                        int lb_scale, int noise_absc, const Q &x) {
  const int bands = nsb - skip;
  const int start_up = cx.uni(st->start_up);
  const int ph0 = cx.uni(st->ph_index), harm0 = cx.uni(st->harm_index);
  const int fb_noise_e0 = start_up ? noise_e : cx.uni(st->filt_buf_noise_e);
  cx.sync(); /* everyone has read the scalars before they are updated */
  XsLv idx;
  idx.fill(0);
  XS_LANES(l, 0, 64) idx.own(l) = off + (l & (XS_PK - 1));
  XsLv gain = v.gain.gather(idx);
  const XsLv noise = v.noise.gather(idx), sine = v.sine.gather(idx);
  XS_LANES(k, 0, bands) { /* (the lower half: band k's state) */
    int16_t g[2] = {xs_m(gain.own(k)), xs_e(gain.own(k))};
    if (start_up) {
      st->filt_buf_me[2 * (skip + k)] = g[0];
      st->filt_buf_me[2 * (skip + k) + 1] = g[1];
      st->filt_buf_noise_m[skip + k] = xs_m(noise.own(k));
    } else {
      xs_equalize_filt_buf(&st->filt_buf_me[2 * (skip + k)], g);
      gain.own(k) = xs_me(g[0], g[1]);
    }
  }
  cx.sync();
  XsLv lo;
  lo.fill(0);
  XS_LANES(l, 0, 64) lo.own(l) = l & (XS_PK - 1);
  gain = gain.gather(lo); /* the equalised gains on both halves */
  const XsLv tone = xs_prefix_nonzero_m(cx, sine, nsb).gather(lo); /* (lanes 0 .. nsb - 1 hold the counts) */
  const XsLv s_prev = sine.shifted(cx, -1), s_next = sine.shifted(cx, 1);
  XS_T(21);
  const int nm1 = nsb - 1;
  int fi0 = !(sb_start & 1); /* freq_inv for harmonic index 1; index 3 negates it */
  fi0 = (fi0 << 1) - 1;
  const int n = s1 > s0 ? s1 - s0 : 0;
  const bool crosses = s0 < 32 && s1 > 32;
  const int n_half = (((n + 1) >> 1) + 1) & ~1; /* slots of the lower half of the wave: even, at least half of them */
  XsLv nl_out, fbn_out;
  nl_out.fill(0);
  fbn_out.fill(0);
  XS_LANES(l, 0, 64) {
    const int h = l / XS_PK, k = l & (XS_PK - 1);
    if (k >= nsb) continue;
    const int16_t gm = xs_m(gain.own(l)), ge = xs_e(gain.own(l));
    const int16_t sl = xs_m(sine.own(l));
    const int16_t sl_prev = k > 0 ? xs_m(s_prev.own(l)) : (int16_t)0;
    const int16_t sl_next = (k + 1 < nsb) ? xs_m(s_next.own(l)) : (int16_t)0;
    const int16_t nl_a = xs_m(noise.own(l));
    const int16_t nl_b = k < bands ? xs_noise_rescale(nl_a, final_e - noise_e) : nl_a;
    const int with_noise = !noise_absc && sl == 0;
    const int few_tones = tone.own(l) <= 16;
    const int32_t sine32 = xs_shl(sl, 16);
    int32_t term1;
    if (k == 0) {
      term1 = fx_mul32x16(XS_FACTOR, sl_next);
      if (fi0 < 0) term1 = -term1;
    } else if (k < nm1) {
      const int32_t add = fx_mul32x16(XS_FACTOR, (int16_t)(sl_prev - sl_next));
      term1 = few_tones ? (((k & 1) ? fi0 : -fi0) < 0 ? -add : add) : 0;
    } else {
      const int32_t tms = fx_mul32x16(XS_FACTOR, sl_prev);
      term1 = few_tones ? (((nm1 & 1) ? fi0 : -fi0) > 0 ? tms : -tms) : 0;
    }
    int32_t edge1 = 0;
    int edge_col = -1;
    if (k == 0) {
      edge1 = fx_mul32x16(XS_FACTOR, sl);
      edge_col = sb_start - 1;
    }
    if (k == nm1 && nm1 > 0 && few_tones && k + sb_start < 62) {
      const int32_t tm2 = fx_mul32x16(XS_FACTOR, sl);
      edge1 = ((nm1 & 1) ? fi0 : -fi0) > 0 ? -tm2 : tm2;
      edge_col = sb_start + k + 1;
    }
    const int sh_a = ge - ((adj_e - input_e) - 1), sh_b = ge - ((final_e - input_e) - 1);
    const int shl_a = sh_a > 0 ? (sh_a & 31) : 0, shr_a = sh_a > 0 ? 0 : ((-sh_a) & 31);
    const int shl_b = sh_b > 0 ? (sh_b & 31) : 0, shr_b = sh_b > 0 ? 0 : ((-sh_b) & 31);
    /* half h walks the slots from s0 + h n_half on; n_half is even, so a step's harmonic index is even on both halves or odd
       on both: which of the two bodies (env_calc.c:1564 / :1617) a step runs is a scalar branch, not both under masks */
    const int first = s0 + h * n_half;
    for (int u = 0; u < n_half; u++) {
      const int sl_i = first + u, j = sl_i - s0;
      if (sl_i >= s1) continue;
      const bool late = sl_i >= 32 && s0 < 32; /* behind the change of the noise exponent */
      const int16_t nl = late ? nl_b : nl_a;
      const int ph = (ph0 + j * nsb) & 511, hi = (harm0 + j) & 3;
      const int16_t rp = rand_hi[ph + 1 + k];
      int32_t val = fx_mul32x16(x(sl_i, sb_start + k), gm);
      val = (int32_t)((uint32_t)val << (sl_i < 32 ? shl_a : shl_b)) >> (sl_i < 32 ? shr_a : shr_b);
      const int32_t noisy = xs_mac16x16_shl_sat(val, rp, nl);
      if (!((harm0 + u) & 1)) { /* (uniform) */
        const int32_t toned = hi == 0 ? fx_add_sat(val, sine32) : fx_sub_sat(val, sine32);
        val = with_noise ? noisy : toned;
      } else {
        val = with_noise ? noisy : val;
        val = fx_add_sat(val, hi == 1 ? term1 : -term1);
        if (edge_col >= 0) {
          int32_t t = edge1;
          const int neg = (hi == 1 ? fi0 : -fi0) < 0;
          if (k == 0) {
            const int ne = late ? final_e : noise_e;
            const int16_t nexp = (int16_t)((ne - 16) - lb_scale);
            t = nexp > 0 ? fx_shl(t, nexp) : fx_shr(t, -nexp);
            x(sl_i, edge_col) = neg ? fx_add_sat(x(sl_i, edge_col), t) : fx_sub_sat(x(sl_i, edge_col), t);
          } else {
            x(sl_i, edge_col) = hi == 1 ? fx_add_sat(x(sl_i, edge_col), t) : fx_sub_sat(x(sl_i, edge_col), t);
          }
        }
      }
      x(sl_i, sb_start + k) = val;
    }
    /* what the walk leaves: the noise filter buffer rescaled at the first slot and where the exponent changes, the noise
       level of the last slot */
    int16_t fbn = st->filt_buf_noise_m[k];
    if (n > 0) fbn = xs_noise_rescale(fbn, fb_noise_e0 - noise_e);
    if (crosses) fbn = xs_noise_rescale(fbn, noise_e - final_e);
    fbn_out.own(l) = fbn;
    nl_out.own(l) = crosses ? nl_b : nl_a;
  }
  cx.sync();
  XS_T(22);
  XS_LANES(k, 0, nsb) st->filt_buf_noise_m[k] = (int16_t)fbn_out.own(k);
  cx.sync();
  XS_LANES(k, 0, bands) {
    st->filt_buf_me[2 * (skip + k)] = xs_m(gain.own(k));
    st->filt_buf_noise_m[skip + k] = (int16_t)nl_out.own(k);
  }
  XS_ONE {
    int ne = noise_e;
    if (crosses) ne = final_e;
    st->start_up = 0;
    st->filt_buf_noise_e = n > 0 ? ne : fb_noise_e0;
    st->ph_index = (int16_t)((ph0 + n * nsb) & 511);
    st->harm_index = (int16_t)((harm0 + n) & 3);
  }
  cx.sync();
}

/* ---- HQ (complex) mode -------------------------------------------------------------------------- */
/* env_calc.c:450 */
FX_HD void xs_erg_to_amplitude_hq(const XsCx &cx, int bands, int16_t noise_e, XsEnv &v) {
  XS_LANES(k, 0, bands) {
    int16_t sn[2] = {xs_m(v.sine.own(k)), xs_e(v.sine.own(k))};
    int16_t g[2] = {xs_m(v.gain.own(k)), xs_e(v.gain.own(k))};
    int16_t nl[2] = {xs_m(v.noise.own(k)), xs_e(v.noise.own(k))};
    xs_mant_exp_sqrt(sn);
    xs_mant_exp_sqrt(g);
    xs_mant_exp_sqrt(nl);
    int shift = (noise_e - nl[1]) - 4;
    if (shift > 0) {
      if (shift > 31) shift = 31;
      nl[0] = (int16_t)(nl[0] >> shift);
    } else {
      if (shift < -31) shift = -31;
      nl[0] = (int16_t)xs_shl(nl[0], -shift);
    }
    v.sine.own(k) = xs_me(sn[0], sn[1]);
    v.gain.own(k) = xs_me(g[0], g[1]);
    v.noise.own(k) = xs_me(nl[0], nl[1]);
  }
}

/* xs_erg_to_amplitude_hq for a pass (see "two envelopes side by side") */
FX_HD void xs_erg_to_amplitude_hq_pk(const XsCx &cx, const XsPass &ps, int bands, XsEnv &v) {
  XS_LANES(l, 0, 64) {
    const int q = l / XS_PK, c = l & (XS_PK - 1);
    if (q < ps.n && c < bands) {
      const int noise_e = (int16_t)xs_qsel(q, ps.noise_e);
      int16_t sn[2] = {xs_m(v.sine.own(l)), xs_e(v.sine.own(l))};
      int16_t g[2] = {xs_m(v.gain.own(l)), xs_e(v.gain.own(l))};
      int16_t nl[2] = {xs_m(v.noise.own(l)), xs_e(v.noise.own(l))};
      xs_mant_exp_sqrt(sn);
      xs_mant_exp_sqrt(g);
      xs_mant_exp_sqrt(nl);
      int shift = (noise_e - nl[1]) - 4;
      if (shift > 0) {
        if (shift > 31) shift = 31;
        nl[0] = (int16_t)(nl[0] >> shift);
      } else {
        if (shift < -31) shift = -31;
        nl[0] = (int16_t)xs_shl(nl[0], -shift);
      }
      v.sine.own(l) = xs_me(sn[0], sn[1]);
      v.gain.own(l) = xs_me(g[0], g[1]);
      v.noise.own(l) = xs_me(nl[0], nl[1]);
    }
  }
}

/* ---- two envelopes side by side, low-power mode ------------------------------------------------------------------------
 * The same arrangement for the low-power chain (env_calc.c:692 with low_pow_flag): the energies (real matrix), the gain
 * mathematics and the limiter are the functions above; the alias reduction (env_calc.c:78) and the amplitude conversion
 * (env_calc.c:423) follow here on the same lanes -- element k of envelope q on lane 32 q + k -- and the slots of the two
 * envelopes are then adjusted one after the other by xs_adapt_noise_gain_lp on the envelope's half of the vectors.  Taken
 * for regular frames (xs_pack_frame_ok) whose adjusted range starts at sub_band_start (the reference indexes its gain
 * arrays from max_qmf_subband_aac and the aliasing degrees from sub_band_start with one counter; with the two apart the
 * one-envelope chain restates what that does). */

/* xs_alias_groups for a pass.  deg1p: element (q, k) = deg1[k].  A run cannot leave its half: the last band of an envelope
   (k = nsb - 1 <= 31) is never in a run. */
FX_HD uint64_t xs_alias_groups_pk(const XsCx &cx, const XsPass &ps, const XsLv &deg1p, const XsLv &alias_red, int nsb,
                                  XsLv &end) {
  XsLv f;
  f.fill(0);
  XS_LANES(l, 0, 64) {
    const int q = l / XS_PK, k = l & (XS_PK - 1);
    if (q < ps.n && k < nsb)
      f.own(l) = ((k < nsb - 1 && deg1p.own(l) != 0 && alias_red.own(l)) ? 1 : 0) | (alias_red.own(l) ? 2 : 0);
  }
  const uint64_t in_run = xs_ballot(cx, f, 1, 64), red = xs_ballot(cx, f, 2, 64);
  XsLv st;
  st.fill(0);
  XS_LANES(l, 0, 64) {
    if (!((in_run >> l) & 1)) continue;
    const int base = l & ~(XS_PK - 1);
    const uint64_t gaps_below = ~in_run & (xs_mask_upto(l) >> 1);
    int run_start = gaps_below ? 64 - xs_clz64(gaps_below) : 0;
    if (run_start < base) run_start = base;
    if ((l - run_start) & 3) continue;
    st.own(l) = 1;
    int e;
    if (((in_run >> l) & 15) == 15) {
      e = l + 4;
    } else {
      const int z = l + xs_ctz64(~(in_run >> l)); /* first lane at or above l outside the run */
      if (z - base >= nsb - 1)
        e = base + nsb;
      else
        e = ((red >> z) & 1) ? z + 1 : z;
    }
    end.own(l) = e;
  }
  return xs_ballot(cx, st, 1, 64);
}
/* xs_alias_reduction for a pass: groups, sums and results by lane number (a group's lanes are neighbours in its half) */
FX_HD void xs_alias_reduction_pk(const XsCx &cx, const XsPass &ps, XsEnv &v, const XsLv &degp, const XsLv &deg1p, XsWork *w,
                                 uint64_t starts, const XsLv &end, int nsb) {
  if (starts == 0) return;
  XS_LANES(l, 0, 64) {
    w->fold_a[l][0] = v.est.own(l);
    w->fold_a[l][1] = v.gain.own(l);
  }
  cx.sync();
  XS_LANES(s, 0, 64) {
    if (!((starts >> s) & 1)) continue;
    int16_t amp_m, amp_e, gg_m, gg_e;
    xs_avggain(w->fold_a, s, end.own(s), &amp_m, &amp_e, &gg_m, &gg_e, 1);
    w->res_a[s][0] = amp_m;
    w->res_a[s][1] = amp_e;
    w->res_a[s][2] = gg_m;
    w->res_a[s][3] = gg_e;
    w->res_b[s][0] = (int16_t)end.own(s);
  }
  cx.sync();
  XsLv mine;
  mine.fill(-1);
  XS_LANES(l, 0, 64) {
    const int q = l / XS_PK, k = l & (XS_PK - 1);
    if (!(q < ps.n && k < nsb)) continue;
    const uint64_t below = starts & xs_mask_upto(l);
    if (!below) continue;
    const int s = 63 - xs_clz64(below);
    if (l >= w->res_b[s][0]) continue; /* (a group of the other half ends below this half: never taken for it) */
    mine.own(l) = s;
    const int16_t gg_m = w->res_a[s][2], gg_e = w->res_a[s][3];
    int16_t alpha = (int16_t)degp.own(l);
    if (k < nsb - 1 && (int16_t)deg1p.own(l) > alpha) alpha = (int16_t)deg1p.own(l);
    int32_t gain_m = (int32_t)alpha * gg_m;
    int16_t one_minus = (int16_t)(0x7fff - alpha);
    int32_t tm = xs_m(v.gain.own(l)), te = xs_e(v.gain.own(l));
    tm = ((int32_t)one_minus * tm) >> 15;
    int32_t d = gg_e - te;
    if (d >= 0) {
      te = gg_e;
      tm = fx_shr(tm, d);
      tm = (gain_m >> 15) + tm;
    } else {
      tm = fx_shr(gain_m, 15 - d) + tm;
    }
    v.gain.own(l) = xs_me((int16_t)tm, (int16_t)te);
    w->fold_b[l][0] = (int32_t)((uint32_t)tm * (uint32_t)(int32_t)xs_m(v.est.own(l))) >> 16;
    w->fold_b[l][1] = te + xs_e(v.est.own(l)) + 1;
  }
  cx.sync();
  XS_LANES(s, 0, 64) {
    if (!((starts >> s) & 1)) continue;
    int32_t mod_m = 0, mod_e = 0;
    const int e = end.own(s);
    for (int k = s; k < e; k++) xs_acc_me(&mod_m, &mod_e, w->fold_b[k][0], w->fold_b[k][1]);
    int nv = 16 - xs_pnorm32(mod_m);
    if (nv > 0) {
      mod_m >>= nv;
      mod_e += nv;
    }
    int16_t comp_m;
    int comp_e = xs_fix_mant_div(w->res_a[s][0], (int16_t)mod_m, &comp_m);
    comp_e = (int16_t)(comp_e + w->res_a[s][1] - (int16_t)mod_e + 1 + 1);
    w->res_b[s][1] = comp_m;
    w->res_a[s][3] = (int16_t)comp_e;
  }
  cx.sync();
  XS_LANES(l, 0, 64) {
    const int s = mine.own(l);
    if (s < 0) continue;
    const int16_t comp_m = w->res_b[s][1], comp_e = w->res_a[s][3];
    const int32_t g2 = v.gain.own(l);
    v.gain.own(l) = xs_me((int16_t)(((int32_t)xs_m(g2) * comp_m) >> 16), (int16_t)(xs_e(g2) + comp_e));
  }
  cx.sync();
}
/* xs_erg_to_amplitude_lp for a pass */
FX_HD void xs_erg_to_amplitude_lp_pk(const XsCx &cx, const XsPass &ps, int bands, XsEnv &v) {
  XS_LANES(l, 0, 64) {
    const int q = l / XS_PK, c = l & (XS_PK - 1);
    if (q < ps.n && c < bands) {
      const int noise_e = (int16_t)xs_qsel(q, ps.noise_e);
      int16_t sn[2] = {xs_m(v.sine.own(l)), xs_e(v.sine.own(l))};
      int16_t g[2] = {xs_m(v.gain.own(l)), xs_e(v.gain.own(l))};
      int16_t nl[2] = {xs_m(v.noise.own(l)), xs_e(v.noise.own(l))};
      xs_mant_exp_sqrt(sn);
      xs_mant_exp_sqrt(g);
      xs_mant_exp_sqrt(nl);
      int shift = (noise_e - nl[1]) - 4;
      if (shift > 0)
        nl[0] = (int16_t)xs_sar(nl[0], shift);
      else
        nl[0] = (int16_t)xs_shl(nl[0], -shift);
      shift = sn[1] - noise_e;
      if (shift > 0)
        sn[0] = xs_shl16_sat(sn[0], (int16_t)shift);
      else
        sn[0] = (int16_t)xs_sar(sn[0], (int16_t)-shift);
      v.sine.own(l) = xs_me(sn[0], sn[1]);
      v.gain.own(l) = xs_me(g[0], g[1]);
      v.noise.own(l) = xs_me(nl[0], nl[1]);
    }
  }
}

/* the regular frame the two-envelope chain is written for: energies per QMF band (interpol_freq), both frequency tables
   spanning exactly the SBR range of at most 32 bands, the limiter table inside it.  Every frame the reference's own tables
   (ixheaacd_freq_sca.c) describe at these widths passes; anything else runs the one-envelope chain. */
FX_HD bool xs_pack_frame_ok(const XsCx &cx, const xaac_sbr_header *h) {
  const int nhi = cx.uni(h->num_sf_bands[1]), nlo = cx.uni(h->num_sf_bands[0]), nlf = cx.uni(h->num_lf_bands);
  const int sb_start = cx.uni(h->sub_band_start), sb_end = cx.uni(h->sub_band_end), nsb = sb_end - sb_start;
  if (!cx.uni(h->interpol_freq) || nhi < 1 || nlo < 1 || nsb < 1 || nsb > XS_PK) return false;
  if (cx.uni(h->freq_band_tbl_hi[0]) != sb_start || cx.uni(h->freq_band_tbl_lo[0]) != sb_start) return false;
  if (cx.uni(h->freq_band_tbl_hi[nhi]) != sb_end || cx.uni(h->freq_band_tbl_lo[nlo]) != sb_end) return false;
  int32_t bad = 0;
  XS_LANES(c, 0, nlf + 1) bad |= h->freq_band_tbl_lim[c] > nsb;
  return cx.wave_or(bad) == 0;
}

/* N slots of one band inside a segment of constant scale (xs_adapt_noise_gain_hq, step 2): the per-slot body of
   ixheaacd_harm_idx_zerotwo / _onethree with the gain, noise level and sine levels of the segment.  ph / harm
   advance by N slots. */
struct XsApplyHq {
  int32_t sl_even, sl_odd;
  int ls, rs, keep, col, step, kk, harm_lane; /* ls / rs / keep: the segment's shift, see xs_adapt_noise_gain_hq */
  int16_t sg, snz;
  bool tone, noise, fi, live;
  int32_t hr, hr_keep; /* OR of the magnitudes written (hr_keep: -1 where they count -- slots below 32 --, else 0) */
};
template <int N, class Q>
FX_HD void xs_apply_slots_hq(const Q &x, XsApplyHq &a, int l, int &ph, int &harm) {
  int32_t rp[N], xr[N], xi[N];
  XS_UNROLL
  for (int j = 0; j < N; j++) rp[j] = XS_TAB_RAND(((ph + j * a.step) & 511) + 1 + a.kk);
  XS_UNROLL
  for (int j = 0; j < N; j++) {
    xr[j] = x(l + j, a.col);
    xi[j] = x.im(l + j, a.col);
  }
  XS_UNROLL
  for (int j = 0; j < N; j++) {
    int32_t re = fx_mul32x16(xr[j], a.sg), im = fx_mul32x16(xi[j], a.sg);
    re = (fx_shlw(re, a.ls) >> a.rs) & a.keep; /* = shift > 0 ? fx_shl(re, shift) : fx_shr(re, -shift) */
    im = (fx_shlw(im, a.ls) >> a.rs) & a.keep;
    /* a.harm_lane is the harmonic index as a per-lane value (the same in every lane): the four cases become lane
       selects instead of scalar branches between the slots */
    const int hi = (a.harm_lane + j) & 3;
    const bool plus = a.fi != (hi == 1);
    const int32_t re_a = fx_add_sat(re, a.sl_even), re_s = fx_sub_sat(re, a.sl_even);
    const int32_t im_a = fx_add_sat(im, a.sl_odd), im_s = fx_sub_sat(im, a.sl_odd);
    const int32_t re_t = hi == 0 ? re_a : (hi == 2 ? re_s : re);
    const int32_t im_t = (hi & 1) ? (plus ? im_a : im_s) : im;
    const int32_t re_n = xs_mac16x16_shl_sat(re, (int16_t)(rp[j] >> 16), a.snz);
    const int32_t im_n = xs_mac16x16_shl_sat(im, (int16_t)rp[j], a.snz);
    xr[j] = a.tone ? re_t : (a.noise ? re_n : re);
    xi[j] = a.tone ? im_t : (a.noise ? im_n : im);
    a.hr |= (fx_abs_nrm(xr[j]) | fx_abs_nrm(xi[j])) & a.hr_keep;
  }
  if (a.live) {
    XS_UNROLL
    for (int j = 0; j < N; j++) {
      x(l + j, a.col) = xr[j];
      x.im(l + j, a.col) = xi[j];
    }
  }
  ph = (ph + N * a.step) & 511;
  harm = (harm + N) & 3;
  a.harm_lane = (a.harm_lane + N) & 3;
}

/* env_calc.c:479 (HQ branch) with ixheaacd_adj_timeslot (env_dec.c:845) and ixheaacd_harm_idx_zerotwo /
   _onethree (env_calc.c:1759 / :1827): gain smoothing over the first slots of an envelope, then gain,
   noise (complex random phase) or sine (real part for harmonic index 0/2, imaginary for 1/3, sign
   alternating with the band) per band.  Bands are independent; lane i owns filter-buffer entry i and,
   from skip on, band i - skip. */
/* The envelope adjuster's memory (ixheaacd_env_calc.h:23-32) while a frame is worked on, HQ mode: the smoothing filter's
   gain and noise entries as lane vectors, the four scalars as wave-uniform values -- loaded once per frame and stored once,
   where every envelope used to go through the state's LDS copy half a dozen times (each a wait in a serial stretch).
   Element li holds entry i = li & 31 when the SBR range fits 32 bands (`two`: both halves of the wave hold the same entries,
   see xs_adapt_noise_gain_hq), else entry li. */
struct XsAdjMem {
  XsLv fb, fn; /* filt_buf_me as (mantissa | exponent << 16), filt_buf_noise_m */
  XsLv hr;     /* OR of the magnitudes the envelopes have written to slots 0..31 so far (env_calc.c:961's headroom, taken on
                  the way instead of by a scan of the matrix afterwards) */
  int start_up, ph_index, harm_index, filt_buf_noise_e;
  bool two;
};
template <class ST>
FX_HD void xs_adj_load(const XsCx &cx, const ST *st, int nsb, XsAdjMem &m) {
  m.two = nsb <= 32;
  m.fb.fill(0);
  m.fn.fill(0);
  m.hr.fill(0);
  XS_LANES(li, 0, 64) {
    const int i = m.two ? li & 31 : li;
    if (i < nsb && i < XS_MAXF) {
      m.fb.own(li) = xs_me(st->filt_buf_me[2 * i], st->filt_buf_me[2 * i + 1]);
      m.fn.own(li) = st->filt_buf_noise_m[i];
    }
  }
  m.start_up = cx.uni(st->start_up);
  m.ph_index = cx.uni(st->ph_index);
  m.harm_index = cx.uni(st->harm_index);
  m.filt_buf_noise_e = cx.uni(st->filt_buf_noise_e);
}
template <class ST>
FX_HD void xs_adj_store(const XsCx &cx, ST *st, int nsb, const XsAdjMem &m) {
  XS_LANES(li, 0, m.two ? 32 : 64) {
    if (li < nsb && li < XS_MAXF) {
      st->filt_buf_me[2 * li] = xs_m(m.fb.own(li));
      st->filt_buf_me[2 * li + 1] = xs_e(m.fb.own(li));
      st->filt_buf_noise_m[li] = (int16_t)m.fn.own(li);
    }
  }
  XS_ONE {
    st->start_up = (int16_t)m.start_up;
    st->ph_index = (int16_t)m.ph_index;
    st->harm_index = (int16_t)m.harm_index;
    st->filt_buf_noise_e = (int16_t)m.filt_buf_noise_e;
  }
  cx.sync();
}

/* env_calc.c:479 (HQ branch) with ixheaacd_adj_timeslot (env_dec.c:845) and ixheaacd_harm_idx_zerotwo /
   _onethree (env_calc.c:1759 / :1827): gain smoothing over the first slots of an envelope, then gain,
   noise (complex random phase) or sine (real part for harmonic index 0/2, imaginary for 1/3, sign
   alternating with the band) per band.  Bands are independent; lane i owns filter-buffer entry i and,
   from skip on, band i - skip.  v: the envelope's gains / noise / sine levels, element lane_off + k = band k (lane_off = 32
   for the second envelope of a pass). */
template <class Q>
FX_HD void xs_adapt_noise_gain_hq(const XsCx &cx, XsAdjMem &m, const XsEnv &v, int lane_off, int noise_e, int nsb, int skip,
                                  int s0, int s1, int input_e, int adj_e, int final_e, int sb_start, int noise_absc,
                                  int smooth_length, const Q &x) {
  const int bands = nsb - skip;
  const int start_up = m.start_up;
  const int ph0 = m.ph_index, harm0 = m.harm_index;
  const int fb_noise_e0 = start_up ? noise_e : m.filt_buf_noise_e;
  /* With at most 32 entries the two halves of the wave share the walk: lane 32 g + i takes half g of every segment's slots
     of entry i (a slot's result depends on its number only -- random phase, harmonic index -- once the smoothed start is
     over; the smoothed slots, a recursion, are walked by both halves alike, and so is everything that goes to the memory:
     the halves' copies stay equal).  Band values seen from the lane that owns entry i = skip + k: one gather each. */
  const bool two = m.two;
  XsLv gain_i, noise_i, sine_i;
  {
    XsLv idx;
    idx.fill(0);
    XS_LANES(li, 0, 64) idx.own(li) = ((two ? li & 31 : li) - skip + lane_off) & 63;
    gain_i = v.gain.gather(idx);
    noise_i = v.noise.gather(idx);
    sine_i = v.sine.gather(idx);
  }
  XS_LANES(li, 0, 64) {
    const int i = two ? li & 31 : li;
    const bool band = i >= skip && i < nsb;
    gain_i.own(li) = band ? gain_i.own(li) : 0;
    noise_i.own(li) = band ? noise_i.own(li) : 0;
    sine_i.own(li) = band ? sine_i.own(li) : 0;
    if (band) { /* env_calc.c:1017: the buffer meets the new gain's exponent (or, at start-up, takes the gain) */
      int16_t g[2] = {xs_m(gain_i.own(li)), xs_e(gain_i.own(li))};
      if (start_up) {
        m.fb.own(li) = gain_i.own(li);
        m.fn.own(li) = xs_m(noise_i.own(li));
      } else {
        int16_t fb[2] = {xs_m(m.fb.own(li)), xs_e(m.fb.own(li))};
        xs_equalize_filt_buf(fb, g);
        m.fb.own(li) = xs_me(fb[0], fb[1]);
        gain_i.own(li) = xs_me(g[0], g[1]);
      }
    }
  }
  XS_T(29);
  XS_T(21);
  XS_LANES(li, 0, two ? 64 : nsb) {
    const int g = two ? li >> 5 : 0, i = two ? li & 31 : li;
    if (i >= nsb) continue;
    const int k = i - skip;
    const int16_t gm = xs_m(gain_i.own(li)), ge = xs_e(gain_i.own(li));
    const int16_t sm = xs_m(sine_i.own(li)), se = xs_e(sine_i.own(li));
    int16_t nl = xs_m(noise_i.own(li));
    int16_t fbm = xs_m(m.fb.own(li)), fbn = (int16_t)m.fn.own(li);
    int ne = noise_e, fbe = fb_noise_e0, ph = ph0, harm = harm0;
    int32_t hr = 0;
    int harm_lane0 = harm0; /* a per-lane value on the GPU (see xs_apply_slots_hq) */
    XS_KEEP(harm_lane0);
    const int col = sb_start + (k >= 0 ? k : 0), kk = k >= 0 ? k : 0;
    /* 1. the envelope's first slots while the gains are smoothed (at most four; the reference's slot body as it is) */
    int l = s0;
    int n_smooth = s1 - s0 < smooth_length ? s1 - s0 : smooth_length;
    for (; n_smooth > 0; n_smooth--, l++) {
      int scale_change;
      if (l < 32) {
        scale_change = adj_e - input_e;
      } else {
        scale_change = final_e - input_e;
        if (l == 32 && s0 < 32) {
          const int diff = final_e - ne;
          ne = final_e;
          if (k >= 0) nl = xs_noise_rescale(nl, diff);
        }
      }
      fbn = xs_noise_rescale(fbn, fbe - ne);
      fbe = ne;
      const int32_t rp = XS_TAB_RAND((ph & 511) + 1 + kk);
      const int hi = harm;
      ph = (ph + bands) & 511;
      harm = (harm + 1) & 3;
      if (k < 0) continue;
      const int16_t smooth = XS_TAB_SMOOTH(l - s0);
      int16_t sg = gm, snz = nl;
      if (smooth) {
        const int16_t direct = fx_sat16(0x7fff - (int32_t)smooth);
        const int16_t t = (int16_t)(xs_mult16(smooth, fbm) + xs_mult16(direct, gm));
        const int16_t t1 = (int16_t)(xs_mult16(smooth, fbn) + xs_mult16(direct, nl));
        fbm = (int16_t)(t << 1);
        fbn = (int16_t)(t1 << 1);
        sg = fbm;
        snz = fbn;
      }
      const int16_t sc = (int16_t)((int16_t)scale_change - 1);
      int32_t re = fx_mul32x16(x(l, col), sg), im = fx_mul32x16(x.im(l, col), sg);
      const int shift = (int16_t)(ge - sc);
      if (shift > 0) {
        re = fx_shl(re, shift);
        im = fx_shl(im, shift);
      } else {
        re = fx_shr(re, -shift);
        im = fx_shr(im, -shift);
      }
      if (sm != 0) {
        const int tmp = (int16_t)(se - (int16_t)(ne - 16));
        int32_t sine_level;
        if (!(hi & 1)) {
          /* (sic) the non-positive case shifts by tmp, not -tmp (env_calc.c:1797) */
          sine_level = tmp > 0 ? fx_shl(sm, tmp) : fx_shr(sm, tmp);
          re = hi == 0 ? fx_add_sat(re, sine_level) : fx_sub_sat(re, sine_level);
        } else {
          sine_level = tmp > 0 ? fx_shl(sm, tmp) : fx_shr(sm, -tmp);
          int fi = sb_start & 1;
          if (hi == 1) fi = !fi;
          if (k & 1) fi = !fi;
          im = fi ? fx_add_sat(im, sine_level) : fx_sub_sat(im, sine_level);
        }
      } else if (!noise_absc) {
        re = xs_mac16x16_shl_sat(re, (int16_t)(rp >> 16), snz);
        im = xs_mac16x16_shl_sat(im, (int16_t)rp, snz);
      }
      hr |= (g == 0 && l < 32) ? (fx_abs_nrm(re) | fx_abs_nrm(im)) : 0; /* (the half that writes the slot) */
      if (g == 0) {
        x(l, col) = re;
        x.im(l, col) = im;
      }
    }
    XS_T(30);
    /* 2. the rest in at most two segments, slots below 32 and from 32 on: inside a segment every per-slot quantity
       but the random phase and the harmonic index is a constant of the band, so the slots go in bursts of eight
       (rows and random phases fetched together, no control flow between the slots) */
    while (l < s1) {
      const int seg_end = (l < 32 && s1 > 32) ? 32 : s1;
      int scale_change;
      if (l < 32) {
        scale_change = adj_e - input_e;
      } else {
        scale_change = final_e - input_e;
        if (l == 32 && s0 < 32) {
          const int diff = final_e - ne;
          ne = final_e;
          if (k >= 0) nl = xs_noise_rescale(nl, diff);
        }
      }
      fbn = xs_noise_rescale(fbn, fbe - ne); /* a no-op in the segment's later slots */
      fbe = ne;
      XsApplyHq a;
      a.sg = gm;
      a.snz = nl;
      {
        /* fx_shl / fx_shr take their count mod 256; a left shift by more than 31 leaves 0, a right one the sign */
        const int shift = (int16_t)(ge - (int16_t)((int16_t)scale_change - 1));
        const int b = (shift > 0 ? shift : -shift) & 0xff;
        a.ls = shift > 0 && b <= 31 ? b : 0;
        a.rs = shift > 0 ? 0 : (b < 31 ? b : 31);
        a.keep = shift > 0 && b > 31 ? 0 : -1;
      }
      const int tmp = (int16_t)(se - (int16_t)(ne - 16));
      a.sl_even = tmp > 0 ? fx_shl(sm, tmp) : fx_shr(sm, tmp); /* (sic), as above */
      a.sl_odd = tmp > 0 ? fx_shl(sm, tmp) : fx_shr(sm, -tmp);
      a.tone = k >= 0 && sm != 0;
      a.noise = k >= 0 && sm == 0 && !noise_absc;
      a.fi = ((sb_start ^ kk) & 1) != 0;
      a.live = k >= 0;
      a.col = col;
      a.step = bands;
      a.kk = kk;
      a.hr = 0;
      a.hr_keep = (k >= 0 && l < 32) ? -1 : 0;
      /* this lane's share of the segment's slots [l, seg_end) */
      const int count = seg_end - l, half = two ? (count + 1) >> 1 : count;
      int lg = g ? l + half : l, phg = g ? (ph + half * bands) & 511 : ph, harmg = g ? (harm + half) & 3 : harm;
      const int end_g = g ? seg_end : l + half;
      a.harm_lane = (harm_lane0 + (lg - s0)) & 3;
      for (; lg + 8 <= end_g; lg += 8) xs_apply_slots_hq<8>(x, a, lg, phg, harmg);
      if (lg + 4 <= end_g) {
        xs_apply_slots_hq<4>(x, a, lg, phg, harmg);
        lg += 4;
      }
      if (lg + 2 <= end_g) {
        xs_apply_slots_hq<2>(x, a, lg, phg, harmg);
        lg += 2;
      }
      if (lg < end_g) {
        xs_apply_slots_hq<1>(x, a, lg, phg, harmg);
        lg += 1;
      }
      hr |= a.hr;
      ph = (ph + count * bands) & 511;
      harm = (harm + count) & 3;
      l = seg_end;
    }
    /* the memory as the envelope leaves it (env_calc.c:1060): a band's entry takes the envelope's gain and noise level */
    m.fb.own(li) = xs_me(k >= 0 ? gm : fbm, xs_e(m.fb.own(li)));
    m.fn.own(li) = k >= 0 ? nl : fbn;
    m.hr.own(li) |= hr;
  }
  cx.sync();
  XS_T(22);
  {
    const int n = s1 > s0 ? s1 - s0 : 0;
    int ne = noise_e;
    if (s0 < 32 && s1 > 32) ne = final_e;
    m.start_up = 0;
    m.filt_buf_noise_e = n > 0 ? ne : fb_noise_e0;
    m.ph_index = (int16_t)((ph0 + n * bands) & 511);
    m.harm_index = (int16_t)((harm0 + n) & 3);
  }
}

/* lpp_tran.c:372: complex covariances of low band k over `slots` (= 38) slots starting at row 0, with the
   two LPC history rows below.  All sums wrap, so their order is free: with M(a, b) = a * hi16(b) >> 16,
   phi_01 = sum_{n=0}^{slots-1} x[n] conj(x[n-1]), phi_02 likewise with x[n-2], phi_11 = sum_{n=-1}^{slots-2}
   |x[n]|^2, phi_12 = sum_{n=-1}^{slots-2} x[n] conj(x[n-1]), phi_22 = sum_{n=-2}^{slots-3} |x[n]|^2. */
struct XsCovHq {
  int32_t phi_11, phi_22, phi_01, phi_02, phi_12, phi_01_im, phi_02_im, phi_12_im;
};
/* the terms of slots [n0, n1) only; `first` adds the terms that precede slot 0 (the caller sums the parts) */
template <class Q>
FX_HD void xs_covariance_hq(const Q &x, int k, int n0, int n1, int slots, bool first, XsCovHq *c) {
  int32_t p01 = 0, p01i = 0, p02 = 0, p02i = 0, p11 = 0, p12 = 0, p12i = 0, p22 = 0;
  int32_t r2 = fx_shr(x(n0 - 2, k), 3), i2 = fx_shr(x.im(n0 - 2, k), 3); /* x[n-2] */
  int32_t r1 = fx_shr(x(n0 - 1, k), 3), i1 = fx_shr(x.im(n0 - 1, k), 3); /* x[n-1] */
  if (first) {
    p22 = fx_add(xs_mul_hi16(r2, r2), xs_mul_hi16(i2, i2));
    p12 = fx_add(xs_mul_hi16(r1, r2), xs_mul_hi16(i1, i2));
    p12i = fx_sub(xs_mul_hi16(i1, r2), xs_mul_hi16(r1, i2));
  }
  XS_UNROLL4
  for (int n = n0; n < n1; n++) {
    const int32_t r0 = fx_shr(x(n, k), 3), i0 = fx_shr(x.im(n, k), 3);
    const int32_t t01 = fx_add(xs_mul_hi16(r0, r1), xs_mul_hi16(i0, i1));
    const int32_t t01i = fx_sub(xs_mul_hi16(i0, r1), xs_mul_hi16(r0, i1));
    const int32_t e1 = fx_add(xs_mul_hi16(r1, r1), xs_mul_hi16(i1, i1)); /* |x[n-1]|^2 */
    p01 = fx_add(p01, t01);
    p01i = fx_add(p01i, t01i);
    p02 = fx_add(p02, fx_add(xs_mul_hi16(r0, r2), xs_mul_hi16(i0, i2)));
    p02i = fx_add(p02i, fx_sub(xs_mul_hi16(i0, r2), xs_mul_hi16(r0, i2)));
    p11 = fx_add(p11, e1);
    if (n < slots - 1) { /* the shifted sums stop one sample earlier */
      p12 = fx_add(p12, t01);
      p12i = fx_add(p12i, t01i);
      p22 = fx_add(p22, e1);
    }
    r2 = r1;
    i2 = i1;
    r1 = r0;
    i1 = i0;
  }
  c->phi_11 = p11;
  c->phi_22 = p22;
  c->phi_01 = p01;
  c->phi_02 = p02;
  c->phi_12 = p12;
  c->phi_01_im = p01i;
  c->phi_02_im = p02i;
  c->phi_12_im = p12i;
}

/* lpp_tran.c:1041-1201: complex prediction coefficients of one low band, with the reference's reset
   rules.  a[0..3] = alpha0 re, alpha0 im, alpha1 re, alpha1 im. */
FX_HD void xs_lpc_coeffs_hq(const XsCovHq *s, int16_t *a) {
  int32_t mx = fx_abs_nrm(s->phi_01) | fx_abs_nrm(s->phi_02) | fx_abs_nrm(s->phi_12) | s->phi_11 | s->phi_22 |
               fx_abs_nrm(s->phi_01_im) | fx_abs_nrm(s->phi_02_im) | fx_abs_nrm(s->phi_12_im);
  const int q = xs_pnorm32(mx);
  const int32_t p11 = xs_shl(s->phi_11, q), p22 = xs_shl(s->phi_22, q), p01 = xs_shl(s->phi_01, q),
                p02 = xs_shl(s->phi_02, q), p12 = xs_shl(s->phi_12, q), p01i = xs_shl(s->phi_01_im, q),
                p02i = xs_shl(s->phi_02_im, q), p12i = xs_shl(s->phi_12_im, q);
  const int32_t d = fx_shlw(fx_sub_sat(fx_mul32(p11, p22), fx_add_sat(fx_mul32(p12, p12), fx_mul32(p12i, p12i))), 1);
  int reset = 0;
  int16_t a1r = 0, a1i = 0, a0r = 0, a0i = 0;
  if (d != 0) {
    const int norm_d = fx_norm32(d);
    const int16_t inv_d = (int16_t)xs_fix_div(0x40000000, xs_shl(d, norm_d));
    const int32_t mod_d = fx_abs_sat(d);
    const int32_t tr = fx_sub_sat(fx_sub_sat(fx_mul32(p01, p12), fx_mul32(p01i, p12i)), fx_mul32(p02, p11)) >> 1;
    const int32_t ti = fx_sub_sat(fx_add_sat(fx_mul32(p01i, p12), fx_mul32(p01, p12i)), fx_mul32(p02i, p11)) >> 1;
    if (fx_abs_sat(tr) >= mod_d)
      reset = 1;
    else
      a1r = (int16_t)(xs_shl(fx_mul32x16(tr, inv_d), norm_d + 1) >> 15);
    if (fx_abs_sat(ti) >= mod_d)
      reset = 1;
    else
      a1i = (int16_t)(xs_shl(fx_mul32x16(ti, inv_d), norm_d + 1) >> 15);
  }
  if (p11 != 0) {
    const int norm = fx_norm32(p11);
    const int16_t inv = (int16_t)xs_fix_div(0x40000000, xs_shl(p11, norm));
    int32_t tr = fx_add_sat(fx_add(p01 >> 3, fx_mul32x16(p12, a1r)), fx_mul32x16(p12i, a1i));
    int32_t ti = fx_sub_sat(fx_add(p01i >> 3, fx_mul32x16(p12, a1i)), fx_mul32x16(p12i, a1r));
    tr = fx_shlw(tr, 1);
    ti = fx_shlw(ti, 1);
    if (fx_abs_sat(tr) >= p11)
      reset = 1;
    else
      a0r = (int16_t)(xs_shl(fx_mul32x16(fx_sub_sat(0, tr), inv), norm + 1) >> 15);
    if (fx_abs_sat(ti) >= p11)
      reset = 1;
    else
      a0i = (int16_t)(xs_shl(fx_mul32x16(fx_sub_sat(0, ti), inv), norm + 1) >> 15);
  }
  if (fx_add_sat((int32_t)a0r * a0r, (int32_t)a0i * a0i) >= 0x40000000) reset = 1;
  if (fx_add_sat((int32_t)a1r * a1r, (int32_t)a1i * a1i) >= 0x40000000) reset = 1;
  if (reset) a0r = a0i = a1r = a1i = 0;
  a[0] = a0r;
  a[1] = a0i;
  a[2] = a1r;
  a[3] = a1i;
}

/* lpp_tran.c:102 + :1203-1250: copy / inverse-filter low band lb into high band hb of patch `patch`.  The reference
   walks the low bands and, inside, the patches; every (low band, patch) pair writes its own high band, so here the
   pairs are spread over the lanes by their high band (xs_hf_generator_hq). */
template <class Q>
FX_HD void xs_patch_band_hq(const xaac_sbr_header *h, const Q &x, int lb, int hb, const int16_t *alpha,
                            const int32_t *bw_array, int start_idx, int stop_idx) {
  /* the reference's per-patch running index: first border above hb, capped (lpp_tran.c:1218).  The walk stops at the first
     border above hb; all the borders it may look at are fetched together (one LDS latency instead of one per step) */
  constexpr int kBwCap = XAAC_SBR_MAX_PATCHES - 1 < XAAC_SBR_MAX_NOISE_VALUES ? XAAC_SBR_MAX_PATCHES - 1 : XAAC_SBR_MAX_NOISE_VALUES;
  int bord[kBwCap];
  XS_UNROLL
  for (int i = 0; i < kBwCap; i++) bord[i] = h->bw_borders[i];
  int bi = 0;
  bool on = true;
  XS_UNROLL
  for (int i = 0; i < kBwCap; i++) {
    on = on && hb >= bord[i];
    bi += on ? 1 : 0;
  }
  int16_t bw = (int16_t)(bw_array[bi] >> 16);
  const int16_t a0r = xs_mult16_shl_sat(bw, alpha[0]), a0i = xs_mult16_shl_sat(bw, alpha[1]);
  bw = xs_mult16_shl_sat(bw, bw);
  const int16_t a1r = xs_mult16_shl_sat(bw, alpha[2]), a1i = xs_mult16_shl_sat(bw, alpha[3]);
  const int n = stop_idx - start_idx;
  /* where the chirp factor is not positive the reference copies x >> 2 (lpp_tran.c:1239-1248): that is this filter with
     all four coefficients at zero -- its sums are zero then --, so every lane walks the one loop */
  const bool filt = bw > 0;
  const int16_t c0r = filt ? a0r : (int16_t)0, c0i = filt ? a0i : (int16_t)0, c1r = filt ? a1r : (int16_t)0,
                c1i = filt ? a1i : (int16_t)0;
  int32_t p2r = x(start_idx - 2, lb), p2i = x.im(start_idx - 2, lb);
  int32_t p1r = x(start_idx - 1, lb), p1i = x.im(start_idx - 1, lb);
  XS_UNROLL4
  for (int i = 0; i < n; i++) {
    const int32_t cr = x(start_idx + i, lb), ci = x.im(start_idx + i, lb);
    int32_t acc = fx_sub(fx_add(fx_sub(fx_mul32x16(p1r, c0r), fx_mul32x16(p1i, c0i)), fx_mul32x16(p2r, c1r)),
                         fx_mul32x16(p2i, c1i));
    x(start_idx + i, hb) = fx_add(cr >> 2, fx_shlw(acc, 1));
    acc = fx_add(fx_add_sat(fx_add_sat(fx_mul32x16(p1r, c0i), fx_mul32x16(p1i, c0r)), fx_mul32x16(p2r, c1i)),
                 fx_mul32x16(p2i, c1r));
    x.im(start_idx + i, hb) = fx_add(ci >> 2, fx_shlw(acc, 1));
    p2r = p1r;
    p2i = p1i;
    p1r = cr;
    p1i = ci;
  }
}

/* lpp_tran.c:956.  Writes bw_array_prev. */
template <class ST, class Q>
FX_HD void xs_hf_generator_hq(const XsCx &cx, const xaac_sbr_header *h, ST *st, const Q &x, XsWork *w,
                              int start_idx, int last_slot_offset, int max_qmf_subband, const int32_t *invf_mode,
                              const int32_t *invf_mode_prev) {
  const int num_patches = cx.uni(h->num_patches);
  const int stop_idx = cx.uni(h->num_columns) + last_slot_offset;
  XS_PAR(i, 0, XAAC_SBR_MAX_PATCHES) w->bw_array[i] = 0;
  cx.sync();
  xs_invfilt_level_emphasis(cx, st->bw_array_prev, h->num_if_bands, invf_mode, invf_mode_prev, w->bw_array);
  const int actual_stop = cx.uni(
      (int16_t)(h->patch[num_patches - 1].dst_start_band + h->patch[num_patches - 1].num_bands_in_patch));
  {
    const int nz = actual_stop < Q::NB ? Q::NB - actual_stop : 0; /* real and imaginary columns side by side */
    XS_PAR(c, 0, 2 * nz) {
      const int col = c < nz ? actual_stop + c : Q::IM + actual_stop + (c - nz);
      XS_UNROLL4
      for (int l = start_idx; l < stop_idx; l++) x(l, col) = 0;
    }
  }
  const int start_patch = cx.uni(h->start_patch), stop_patch = cx.uni(h->stop_patch);
  XS_PAR(k, start_patch, stop_patch) {
    x(-2, k) = st->lpc_real[0][k];
    x(-1, k) = st->lpc_real[1][k];
    x.im(-2, k) = st->lpc_imag[0][k];
    x.im(-1, k) = st->lpc_imag[1][k];
  }
  cx.sync();
  XS_T(12);
  XsLv al01, al23, src;
  al01.fill(0);
  al23.fill(0);
  /* covariances over num_columns + 6 = 38 slots (lpp_tran.c:1034).  All eight sums wrap, so the slots are dealt out
     to the 64 / w lane groups of w lanes (lane = group * w + low band) and the groups' parts added up. */
  /* (low-delay SBR: over the frame's own num_columns = 16 or 15 slots, lpp_tran.c:1040-1052) */
  const int ncov = Q::LD ? cx.uni(h->num_columns) : 38;
  const int cw = stop_patch <= 16 ? 16 : (stop_patch <= 32 ? 32 : 64), cper = (ncov * cw + 63) / 64;
  XsLv cv[8];
  for (int i = 0; i < 8; i++) cv[i].fill(0);
  XS_LANES(c, 0, 64) {
    const int lb = c & (cw - 1), g = c / cw;
    const int n0 = g * cper, n1 = n0 + cper < ncov ? n0 + cper : ncov;
    if (lb >= start_patch && lb < stop_patch && n0 < n1) {
      XsCovHq part;
      xs_covariance_hq(x, lb, n0, n1, ncov, g == 0, &part);
      cv[0].own(c) = part.phi_11;
      cv[1].own(c) = part.phi_22;
      cv[2].own(c) = part.phi_01;
      cv[3].own(c) = part.phi_02;
      cv[4].own(c) = part.phi_12;
      cv[5].own(c) = part.phi_01_im;
      cv[6].own(c) = part.phi_02_im;
      cv[7].own(c) = part.phi_12_im;
    }
  }
  if (cw < 64)
    for (int i = 0; i < 8; i++) cv[i] = cv[i].fold(cw);
  XS_T(31);
  XS_LANES(lb, start_patch, stop_patch) {
    XsCovHq c;
    c.phi_11 = cv[0].own(lb);
    c.phi_22 = cv[1].own(lb);
    c.phi_01 = cv[2].own(lb);
    c.phi_02 = cv[3].own(lb);
    c.phi_12 = cv[4].own(lb);
    c.phi_01_im = cv[5].own(lb);
    c.phi_02_im = cv[6].own(lb);
    c.phi_12_im = cv[7].own(lb);
    int16_t alpha[4];
    xs_lpc_coeffs_hq(&c, alpha);
    al01.own(lb) = xs_me(alpha[0], alpha[1]);
    al23.own(lb) = xs_me(alpha[2], alpha[3]);
  }
  XS_T(13);
  /* the low band behind each high band: the reference's loops (low bands outside, patches inside) leave the pair with
     the largest low band, the later patch among equals; target bands past 63 are refused by xs_side_info_bad */
  XS_LANES(hb, 0, 64) {
    int best = -1;
    for (int patch = 0; patch < num_patches; patch++) {
      const xaac_sbr_patch *pp = &h->patch[patch];
      const int lb = hb - pp->dst_end_band;
      if (lb < pp->src_start_band || lb >= pp->src_end_band || lb < start_patch || lb >= stop_patch) continue;
      if (hb < max_qmf_subband) continue;
      if (lb >= best) best = lb;
    }
    src.own(hb) = best;
  }
  /* The patch filter is a FIR over the low band's column (it reads x[n-1], x[n-2] of the SOURCE), so a high band's slots can
     be dealt out: when the patched bands fit 32 lanes (they lie at or above max_qmf_subband), lane 32 g + c takes half g of
     band max_qmf_subband + c's slots -- twice the lanes busy, half the walk. */
  int32_t wide = 0;
  XS_LANES(hb, 0, 64) wide |= src.own(hb) >= 0 && (hb < max_qmf_subband || hb >= max_qmf_subband + 32);
  if (cx.wave_or(wide) == 0 && max_qmf_subband >= 0 && stop_idx - start_idx >= 8) {
    XsLv idx;
    idx.fill(0);
    XS_LANES(l, 0, 64) idx.own(l) = (max_qmf_subband + (l & 31)) & 63;
    const XsLv src2 = src.gather(idx);
    XsLv srcc;
    srcc.fill(0);
    XS_LANES(l, 0, 64) srcc.own(l) = (max_qmf_subband + (l & 31) < 64) ? src2.own(l) : -1;
    const XsLv b01 = al01.gather(srcc), b23 = al23.gather(srcc);
    const int mid = start_idx + ((stop_idx - start_idx + 1) >> 1);
    XS_LANES(l, 0, 64) {
      const int lb = srcc.own(l), hb = max_qmf_subband + (l & 31);
      if (lb >= 0) {
        const int16_t alpha[4] = {xs_m(b01.own(l)), xs_e(b01.own(l)), xs_m(b23.own(l)), xs_e(b23.own(l))};
        xs_patch_band_hq(h, x, lb, hb, alpha, w->bw_array, l < 32 ? start_idx : mid, l < 32 ? mid : stop_idx);
      }
    }
  } else {
    const XsLv a01 = al01.gather(src), a23 = al23.gather(src);
    XS_LANES(hb, 0, 64) {
      const int lb = src.own(hb);
      if (lb >= 0) {
        const int16_t alpha[4] = {xs_m(a01.own(hb)), xs_e(a01.own(hb)), xs_m(a23.own(hb)), xs_e(a23.own(hb))};
        xs_patch_band_hq(h, x, lb, hb, alpha, w->bw_array, start_idx, stop_idx);
      }
    }
  }
  cx.sync();
  XS_T(14);
  XS_PAR(i, 0, h->num_if_bands) st->bw_array_prev[i] = w->bw_array[i];
  cx.sync();
}

/* env_calc.c:975-1003: the two shifts that bring the adjusted range to one exponent -- bands [b0, b1): slots [0, first_start)
   by sh_ov, slots [first_start, 32) by sh_main (xs_adjust's meaning).  A caller that is about to move the matrix anyway (the
   GPU core kernel's copy-out) asks for them instead of having them applied in place. */
struct XsPendingAdjust {
  int b0, b1, first_start, sh_ov, sh_main;
};
/* env_calc.c:692, AAC-LC/HE-AAC (not ELD), 1024-sample frames, low-power (Q = XsQmf) or HQ (XsQmfHq).
   deg64: aliasing degree per QMF band from the low-power HF generator.  Returns 0 or -1. */
template <class ST, class Q>
/* env_sf_all / noise_floor_all: the frame's int_env_sf_arr and int_noise_floor.  They are handed in beside `f` so that
   the GPU core keeps only the head of the frame struct in LDS (the 896-byte envelope array is read once per
   envelope: it stays in global memory there) -- nothing below may reach them through `f`. */
FX_HD int xs_calc_sbrenvelope(const XsCx &cx, const xaac_sbr_header *h, const xaac_sbr_frame *f,
                              const int16_t *env_sf_all, const int16_t *noise_floor_all, ST *st, const Q &x,
                              XsWork *w, const int16_t *rand_hi, const XsLv &deg64, XsPendingAdjust *pend = nullptr,
                              const int32_t *sf_words = nullptr) {
  const int num_env = cx.uni(f->num_env);
  const int16_t *border = f->border_vec;
  const int16_t *noise_floor = noise_floor_all;
  const int sb_start = cx.uni(h->sub_band_start), sb_end = cx.uni(h->sub_band_end);
  const int max_sb = cx.uni(f->max_qmf_subband_aac);
  const int nsb = sb_end - sb_start;
  const int skip = max_sb - sb_start;
  const int transient_env = cx.uni(f->transient_env);
  XsEnv v;
  v.est.fill(0);
  v.e_orig.fill(0);
  v.gain.fill(0);
  v.noise.fill(0);
  v.sine.fill(0);
  v.meta.fill(0);
  v.alias_red.fill(0);
  v.deg = deg64.shifted(cx, sb_start);
  v.deg1 = deg64.shifted(cx, sb_start + 1);
  xs_map_sineflags(cx, h->freq_band_tbl_hi, cx.uni(h->num_sf_bands[1]), f->add_harmonics, st, transient_env,
                   v.sine_mapped);
  int adj_e;
  {
    const int prev_sb = cx.uni(st->prev_max_qmf_subband_aac);
    const int first_band = (prev_sb > max_sb ? prev_sb : max_sb) - sb_start;
    int32_t max_noise = 0;
    XS_PAR(i, first_band, nsb)
      if (st->filt_buf_noise_m[i] > max_noise) max_noise = st->filt_buf_noise_m[i];
    max_noise = cx.wave_max(max_noise);
    adj_e = (cx.uni(st->filt_buf_noise_e) - fx_norm32((int16_t)max_noise)) - 16;
  }
  /* the envelopes' scale factors as lane vectors (element j = scale-factor band j; at most 56 per envelope): on the
     GPU the frame's array stays in global memory, and fetching every envelope's run here puts all the loads in flight
     together instead of paying a memory latency per envelope, twice */
  XsLv sfv[XAAC_SBR_MAX_ENVELOPES];
  {
    int base = 0;
    XS_UNROLL
    for (int i = 0; i < XAAC_SBR_MAX_ENVELOPES; i++) {
      sfv[i].fill(0);
      if (i < num_env) {
        const int nsf = cx.uni(h->num_sf_bands[f->freq_res[i]]);
#if defined(__HIP_DEVICE_COMPILE__)
        if (sf_words) {
          /* the GPU core kernel fetched the whole array with the channel-frame's other loads (sbr_core_kernel.hip: r_sf): word
             k of it is in lane k % 64 of sf_words[k / 64]; element base + j comes out of those registers by a lane gather --
             no memory round trip in the middle of the frame */
          const int e = base + cx.lane, k = e >> 1;
          int32_t wv = 0;
          XS_UNROLL
          for (int q = 0; q < (int)((sizeof(((xaac_sbr_frame *)0)->int_env_sf_arr) / 4 + 63) / 64); q++) {
            const int32_t t = __shfl(sf_words[q], k & 63);
            wv = (k >> 6) == q ? t : wv;
          }
          const int16_t hv = (int16_t)((e & 1) ? (wv >> 16) : wv);
          XS_LANES(j, 0, nsf) sfv[i].own(j) = hv;
        } else
#endif
        {
          XS_LANES(j, 0, nsf) sfv[i].own(j) = env_sf_all[base + j];
        }
        base += nsf;
      }
    }
  }
  int final_e = 0;
  /* the frame's grid: two QMF slots per time slot and rows up to MAX_ENV_COLS = 38, or -- low-delay SBR, env_calc.c:847-849 --
     one, inside the frame's own rows; the time slot at which an envelope counts for the next frame's exponent: env_calc.c:811-837 */
  const int t_step = Q::LD ? 1 : 2, nts_hdr = cx.uni(h->num_time_slots);
  const int row_limit = Q::LD ? nts_hdr : 38, frame_end_slot = (Q::LD && nts_hdr == 15) ? 15 : 16;
  {
    XS_UNROLL
    for (int i = 0; i < XAAC_SBR_MAX_ENVELOPES; i++) {
      if (i >= num_env) break;
      int32_t mx = 16 - 16; /* NRG_EXP_OFFSET - SHORT_BITS */
      const int nsf = cx.uni(h->num_sf_bands[f->freq_res[i]]);
      XS_LANES(j, 0, nsf) {
        int t = sfv[i].own(j) & 63;
        if (t > mx) mx = t;
      }
      mx = cx.wave_max(mx);
      mx -= 16;
      int t = (mx + 13) >> 1;
      if (cx.uni(border[i]) < frame_end_slot && t > adj_e) adj_e = (int16_t)t;
      if (cx.uni(border[i + 1]) > frame_end_slot && t > final_e) final_e = (int16_t)t;
    }
  }
  cx.sync();
  XS_T(3);
  int nf_idx = 0;
  const int tansient_env_prev = cx.uni(st->tansient_env_prev);
  const int hb_scale = cx.uni(st->hb_scale), lb_scale = cx.uni(st->lb_scale);
  const XsLv lim_of = xs_limiter_band_of(cx, h, skip);
  XsAdjMem adj; /* HQ: the adjuster's memory in registers from here to the end of the envelopes */
  if constexpr (Q::HQ) xs_adj_load(cx, st, nsb, adj);
  bool packed = false;
#ifndef XS_NO_ENV_PAIRS /* (a checker build runs every frame through the one-envelope chain: tests/test_env_pairs_cpu.py) */
  packed = !Q::LD && xs_pack_frame_ok(cx, h) && (Q::HQ || skip == 0);
#endif
  if (packed) {
    /* two envelopes per pass of the gain mathematics (see "two envelopes side by side") */
    XsLv jn_hi, jn_lo;
    XS_T(8);
    xs_band_maps(cx, h, nsb, jn_hi, jn_lo);
    XS_T(9);
    const int bands = nsb - skip, input_e = 15 - hb_scale, nnf = cx.uni(h->num_nf_bands);
    const int16_t *lim_tab = &XS_TAB_LIMG(2 * cx.uni(h->limiter_gains));
    const int smooth_len_on = (1 - cx.uni(h->smoothing_mode)) << 2;
    int nf_off = 0;
    /* the frame's borders, resolutions and noise borders as lane vectors: a scalar of the pass loop is then a v_readlane,
       not an LDS round trip (with the wave stalled on it) each */
    XsLv bordv, resv, nbordv;
    bordv.fill(0);
    resv.fill(0);
    nbordv.fill(0);
    XS_LANES(j, 0, XAAC_SBR_MAX_ENVELOPES + 1) bordv.own(j) = border[j];
    XS_LANES(j, 0, XAAC_SBR_MAX_ENVELOPES) resv.own(j) = f->freq_res[j];
    XS_LANES(j, 0, XAAC_SBR_MAX_NOISE_ENVELOPES + 1) nbordv.own(j) = f->noise_border_vec[j];
    XsLv degp, deg1p; /* low-power passes: the aliasing degrees on both halves */
    {
      XsLv idx;
      idx.fill(0);
      XS_LANES(l, 0, 64) idx.own(l) = l & (XS_PK - 1);
      degp = v.deg.gather(idx);
      deg1p = v.deg1.gather(idx);
    }
    for (int i = 0; i < num_env;) {
      XsPass ps;
      int s0[2], s1[2];
      /* envelope i: the reference's checks, in its order */
      s0[0] = 2 * bordv.get(i);
      s1[0] = 2 * bordv.get(i + 1);
      if (s0[0] >= 38 || s1[0] > 38 || nf_idx >= XAAC_SBR_MAX_NOISE_ENVELOPES) {
        if constexpr (Q::HQ) xs_adj_store(cx, st, nsb, adj);
        return -1;
      }
      if (bordv.get(i) == nbordv.get(nf_idx + 1)) {
        nf_off += nnf;
        nf_idx++;
      }
      ps.n = 1;
      ps.env[0] = i;
      ps.fr[0] = resv.get(i);
      ps.nf_off[0] = nf_off;
      ps.noise_absc[0] = (i == transient_env || i == tansient_env_prev) ? 1 : 0;
      ps.noise_e[0] = (int16_t)(s0[0] < 32 ? adj_e : final_e);
      ps.env[1] = ps.fr[1] = ps.nf_off[1] = ps.noise_absc[1] = ps.noise_e[1] = 0;
      s0[1] = s1[1] = 0;
      /* envelope i + 1 rides along if it is a regular one behind envelope i: it would pass the same checks, and its slots lie
         behind i's (its energies must not see i's adjusted slots before the reference would) */
      if (i + 1 < num_env) {
        const int t0 = 2 * bordv.get(i + 1), t1 = 2 * bordv.get(i + 2);
        int nf2 = nf_idx, off2 = nf_off;
        bool ok = !(t0 >= 38 || t1 > 38) && nf2 < XAAC_SBR_MAX_NOISE_ENVELOPES && s0[0] < s1[0] && s1[0] <= t0 && t0 < t1;
        if (ok) {
          if (bordv.get(i + 1) == nbordv.get(nf2 + 1)) {
            off2 += nnf;
            nf2++;
          }
          ps.n = 2;
          s0[1] = t0;
          s1[1] = t1;
          ps.env[1] = i + 1;
          ps.fr[1] = resv.get(i + 1);
          ps.nf_off[1] = off2;
          ps.noise_absc[1] = (i + 1 == transient_env || i + 1 == tansient_env_prev) ? 1 : 0;
          ps.noise_e[1] = (int16_t)(t0 < 32 ? adj_e : final_e);
          nf_idx = nf2;
          nf_off = off2;
        }
      }
      /* energies: each envelope's own slots, side by side in v.est */
      XS_T(23);
      xs_energy_per_subband_pk(cx, x, ps, s0, s1, max_sb, bands, input_e, v.est);
      XS_T(4);
      xs_subband_gain_meta_pk(cx, ps, jn_hi, jn_lo, nsb, skip, v);
      XS_T(5);
      xs_calc_subband_gains_pk(cx, ps, xs_pick_env(sfv, ps.env[0]), xs_pick_env(sfv, ps.env[1]), noise_floor_all, bands, skip, v);
      XS_T(6);
      xs_noiselimiting_pk(cx, h, ps, skip, bands, v, w, lim_tab, lim_of);
      XS_T(7);
      if constexpr (Q::HQ) {
        xs_erg_to_amplitude_hq_pk(cx, ps, bands, v);
        XS_T(28);
        /* the envelopes' slots, one envelope after the other */
        for (int q = 0; q < ps.n; q++) {
          /* (the pass's per-envelope values by selects: indexed by the run-time q the arrays lived in scratch memory, and
             the two loads were two exposed memory latencies per envelope) */
          const int absc_q = xs_qsel(q, ps.noise_absc), noise_e_q = xs_qsel(q, ps.noise_e);
          const int smooth_length = absc_q ? 0 : smooth_len_on;
          xs_adapt_noise_gain_hq(cx, adj, v, q ? XS_PK : 0, noise_e_q, nsb, skip, xs_qsel(q, s0), xs_qsel(q, s1), input_e,
                                 adj_e, final_e, max_sb, absc_q, smooth_length, x);
        }
      } else {
        XsLv grp_end;
        grp_end.fill(0);
        const uint64_t grp_starts = xs_alias_groups_pk(cx, ps, deg1p, v.alias_red, nsb, grp_end);
        XS_T(23);
        xs_alias_reduction_pk(cx, ps, v, degp, deg1p, w, grp_starts, grp_end, nsb);
        XS_T(8);
        xs_erg_to_amplitude_lp_pk(cx, ps, bands, v);
        XS_T(9);
        /* the envelopes' slots, one envelope after the other, every second slot on each half of the wave */
        for (int q = 0; q < ps.n; q++)
          xs_adapt_noise_gain_lp_split(cx, st, v, q ? XS_PK : 0, rand_hi, xs_qsel(q, ps.noise_e), nsb, skip, xs_qsel(q, s0),
                                       xs_qsel(q, s1), input_e, adj_e, final_e, max_sb, (int16_t)(15 - lb_scale),
                                       xs_qsel(q, ps.noise_absc), x);
      }
      XS_T(10);
      i += ps.n;
    }
  }
  for (int i = packed ? num_env : 0; i < num_env; i++) {
    const int s0 = t_step * cx.uni(border[i]), s1 = t_step * cx.uni(border[i + 1]);
    if (s0 >= row_limit || s1 > row_limit || nf_idx >= XAAC_SBR_MAX_NOISE_ENVELOPES) {
      if constexpr (Q::HQ) xs_adj_store(cx, st, nsb, adj);
      return -1;
    }
    const int fr = cx.uni(f->freq_res[i]);
    const int16_t *tbl = fr ? h->freq_band_tbl_hi : h->freq_band_tbl_lo;
    const int nsf = cx.uni(h->num_sf_bands[fr]);
    if (cx.uni(border[i]) == cx.uni(f->noise_border_vec[nf_idx + 1])) {
      noise_floor += cx.uni(h->num_nf_bands);
      nf_idx++;
    }
    const int noise_absc = (i == transient_env || i == tansient_env_prev) ? 1 : 0;
    const int smooth_length = noise_absc ? 0 : ((1 - cx.uni(h->smoothing_mode)) << 2);
    const int input_e = 15 - hb_scale;
    if (cx.uni(h->interpol_freq))
      xs_energy_per_subband(cx, x, s0, s1, max_sb, sb_end, input_e, v.est);
    else
      xs_energy_per_sfb(cx, x, nsf, tbl, s0, s1, max_sb, input_e, w, v.est);
    XS_T(4);
    if (cx.uni(tbl[0]) < sb_start) {
      if constexpr (Q::HQ) xs_adj_store(cx, st, nsb, adj);
      return -1;
    }
    const int n_meta = xs_subband_gain_meta(cx, h, max_sb, tbl, nsf, i, v);
    XS_T(5);
    xs_calc_subband_gains(cx, xs_pick_env(sfv, i), noise_floor, i, n_meta, skip, v, noise_absc);
    XS_T(6);
    xs_noiselimiting(cx, h, skip, n_meta, v, w, &XS_TAB_LIMG(2 * cx.uni(h->limiter_gains)), noise_absc, lim_of);
    XS_T(7);
    const int16_t noise_e = (int16_t)(s0 < 32 ? adj_e : final_e);
    if constexpr (!Q::HQ) {
      XsLv grp_end;
      grp_end.fill(0);
      const uint64_t grp_starts = xs_alias_groups(cx, v.deg1, v.alias_red, nsb, grp_end);
      XS_T(23);
      xs_alias_reduction(cx, v, w, grp_starts, grp_end, nsb);
      XS_T(8);
      xs_erg_to_amplitude_lp(cx, nsb - skip, noise_e, v);
      XS_T(9);
      xs_adapt_noise_gain_lp(cx, st, v, rand_hi, noise_e, nsb, skip, s0, s1, input_e, adj_e, final_e, max_sb,
                             (int16_t)(15 - lb_scale), noise_absc, x);
    } else {
      xs_erg_to_amplitude_hq(cx, nsb - skip, noise_e, v);
      xs_adapt_noise_gain_hq(cx, adj, v, 0, noise_e, nsb, skip, s0, s1, input_e, adj_e, final_e, max_sb, noise_absc,
                             smooth_length, x);
    }
    XS_T(10);
  }
  if constexpr (Q::HQ) xs_adj_store(cx, st, nsb, adj);
  const int first_start = cx.uni(border[0]) * 2;
  const int ov_adj_e = 15 - cx.uni(st->ov_hb_scale);
  int ov_reserve = 0, reserve = 0; /* env_calc.c:961: only taken for parametric stereo (never in low-delay SBR, :962) */
  if (!Q::LD && cx.uni(h->channel_mode) == 3) {
    ov_reserve = xs_headroom(cx, x, max_sb, sb_end, 0, first_start);
    /* the envelopes tile [first_start, 32) (the parser's grids do): every word of the range has been written by the slot
       walks, which kept the OR of the magnitudes (XsAdjMem::hr); else the scan */
    bool tiled = Q::HQ && num_env > 0 && 2 * cx.uni(border[num_env]) >= 32;
    for (int i = 0; i < num_env; i++) tiled = tiled && cx.uni(border[i]) < cx.uni(border[i + 1]);
    if (tiled)
      reserve = xs_pnorm32(xs_lv_or(cx, adj.hr) | 1);
    else
      reserve = xs_headroom(cx, x, max_sb, sb_end, first_start, 32);
  }
  const int output_e = (ov_adj_e - ov_reserve) > (adj_e - reserve) ? (ov_adj_e - ov_reserve) : (adj_e - reserve);
  if (pend) {
    pend->b0 = max_sb;
    pend->b1 = sb_end;
    pend->first_start = first_start;
    pend->sh_ov = ov_adj_e - output_e;
    pend->sh_main = adj_e - output_e;
  } else {
    xs_adjust(cx, x, max_sb, sb_end, 0, first_start, ov_adj_e - output_e);
    xs_adjust(cx, x, max_sb, sb_end, first_start, cx.uni(h->num_time_slots) * cx.uni(h->time_step), adj_e - output_e);
  }
  cx.sync();
  XS_ONE {
    st->hb_scale = (int16_t)(15 - output_e);
    st->ov_hb_scale = (int16_t)(15 - final_e);
    st->tansient_env_prev = (transient_env == num_env) ? 0 : -1;
  }
  cx.sync();
  XS_T(11);
  return 0;
}

/* sbrdec_lpfuncs.c:453 (real-valued) */
template <class ST, class Q>
FX_HD void xs_rescale_x_overlap(const XsCx &cx, const xaac_sbr_header *h, const xaac_sbr_frame *f, ST *st,
                                const Q &x) {
  const int old_lsb = cx.uni(st->prev_max_qmf_subband_aac);
  const int start_slot = cx.uni(h->time_step) * (cx.uni(st->prev_end_position) - cx.uni(h->num_time_slots));
  const int new_lsb = cx.uni(f->max_qmf_subband_aac);
  const int ov_hb = cx.uni(st->ov_hb_scale), ov_lb = cx.uni(st->ov_lb_scale), syn_usb = cx.uni(st->syn_usb);
  cx.sync();
  XS_ONE {
    st->codec_usb = (int16_t)new_lsb;
    st->syn_lsb = (int16_t)new_lsb;
  }
  int b0 = old_lsb < new_lsb ? old_lsb : new_lsb, b1 = old_lsb < new_lsb ? new_lsb : old_lsb;
  if (new_lsb == old_lsb || old_lsb <= 0) {
    cx.sync();
    return;
  }
  /* (a previous frame that ended before slot 16 -- no parser's grid does -- gives a negative start: the reference then clears
     rows in front of its buffer; the rows that exist are cleared) */
  /* (low-delay SBR: the rows are this frame's own first slots, cleared before the analysis bank fills them -- its 32 bands -- and
     before sbr_dec.c:1121 zeroes the rest: nothing of the clearing survives, and here the bank has already run) */
  if (!Q::LD) xs_clear(cx, x, old_lsb, new_lsb, start_slot < 0 ? 0 : start_slot, 6);
  int source, target, t_lsb, t_usb;
  if (new_lsb > old_lsb) {
    source = ov_hb;
    target = ov_lb;
    t_lsb = 0;
    t_usb = old_lsb;
  } else {
    source = ov_lb;
    target = ov_hb;
    t_lsb = old_lsb;
    t_usb = syn_usb;
  }
  cx.sync();
  const int reserve = xs_headroom(cx, x, b0, b1, 0, start_slot);
  xs_adjust(cx, x, b0, b1, 0, start_slot, reserve);
  source += reserve;
  int delta = target - source;
  if (delta > 0) {
    delta = -delta;
    b0 = t_lsb;
    b1 = t_usb;
    XS_ONE {
      if (new_lsb > old_lsb)
        st->ov_lb_scale = (int16_t)source;
      else
        st->ov_hb_scale = (int16_t)source;
    }
  }
  cx.sync();
  xs_adjust(cx, x, b0, b1, 0, start_slot, delta);
  cx.sync();
}

/* Side info the code can index with: every count within the capacity of its array, every band number within the
   64-band grid.  The reference's parser guarantees (much tighter) ranges before ixheaacd_sbr_dec ever runs
   (ixheaacd_env_extr.c, ixheaacd_freq_sca.c); the boundary takes the structs from a host it does not control, so a
   frame outside these bounds is refused with -1 like the grid checks of sbr_dec.c:733-748 instead of being indexed.
   A frame the reference decodes always passes. */
template <class ST>
FX_HD int xs_side_info_bad(const XsCx &cx, const xaac_sbr_header *h, const xaac_sbr_frame *f, const ST *st, int ld = 0) {
  int bad = 0;
  /* A frame without SBR processing (apply_processing 0: the decoder has no valid SBR header yet, or lost it) is up-sampled
     by the two banks alone (xs_sbr_core_tail): header and frame side info are whatever the parser last held -- the
     reference's own front end leaves a band range of 32..64 beside a band limit of 15 there -- and nothing below reads
     them, so only the state members that still become band numbers are looked at. */
  const int apply = cx.uni(f->apply_processing) != 0;
  XS_ONE {
    const auto in = [](int v, int lo, int hi) -> int { return v < lo || v > hi; };
    bad |= in(st->prev_end_position, 0, 19) | in(st->prev_max_qmf_subband_aac, 0, 64) | in(st->codec_usb, 0, 64);
    bad |= in(st->syn_lsb, 0, 64) | in(st->syn_usb, 0, 64);
  }
  if (!apply) return cx.wave_or(bad) != 0;
  XS_ONE {
    const auto in = [](int v, int lo, int hi) -> int { return v < lo || v > hi; };
    /* the frame grid this implementation (and the reference's MAX_ENV_COLS buffers, sbr_dec.c:733-748) is laid out
       for: 16 time slots of 2 QMF slots, 32 columns -- decided here, before xs_rescale_x_overlap turns
       time_step * (prev_end_position - num_time_slots) into a row index */
    if (!ld) bad |= (h->num_time_slots != 16) | (h->time_step != 2) | (h->num_columns != 32);
    else bad |= (h->num_time_slots != 16 && h->num_time_slots != 15) | (h->time_step != 1) | (h->num_columns != h->num_time_slots); /* low-delay SBR */
    /* (the state members that become row / band indices are looked at above: a state is the host's to initialise,
       sbrdec_initfuncs.c) */
    bad |= in(h->num_sf_bands[0], 0, XAAC_SBR_MAX_FREQ_COEFFS / 2) | in(h->num_sf_bands[1], 0, XAAC_SBR_MAX_FREQ_COEFFS);
    bad |= in(h->num_nf_bands, 0, XAAC_SBR_MAX_NOISE_COEFFS) | in(h->num_lf_bands, 0, XAAC_SBR_MAX_LIMITERS);
    bad |= in(h->num_if_bands, 0, XAAC_SBR_MAX_NOISE_VALUES) | in(h->limiter_gains, 0, 3);
    bad |= in(h->sub_band_start, 0, 64) | in(h->sub_band_end, h->sub_band_start, 64);
    bad |= in(h->num_patches, 0, XAAC_SBR_MAX_PATCHES) | in(h->start_patch, 0, 64) | in(h->stop_patch, 0, 64);
    bad |= in(f->num_env, 0, XAAC_SBR_MAX_ENVELOPES) | in(f->num_noise_env, 0, XAAC_SBR_MAX_NOISE_ENVELOPES);
    /* the reference sets max_qmf_subband_aac = sub_band_start (sbrdecoder.c:731); below it, skip_bands of env_calc.c:633 /
       :764 would be negative and the per-band arrays would be read before their first element */
    bad |= in(f->max_qmf_subband_aac, h->sub_band_start, 64) | in(f->transient_env, -1, XAAC_SBR_MAX_ENVELOPES);
  }
  /* the entries in use (what lies behind a count is the host's business) */
  const int n_hi = h->num_sf_bands[1], n_lo = h->num_sf_bands[0], n_lim = h->num_lf_bands, n_nf = h->num_nf_bands;
  const int n_if = h->num_if_bands, n_pat = h->num_patches, n_env = f->num_env, n_nenv = f->num_noise_env;
  XS_PAR(i, 0, XAAC_SBR_MAX_FREQ_COEFFS + 1) {
    if (i <= n_hi) bad |= (unsigned)h->freq_band_tbl_hi[i] > 64u;
    if (i <= n_lo && i <= XAAC_SBR_MAX_FREQ_COEFFS / 2) bad |= (unsigned)h->freq_band_tbl_lo[i] > 64u;
    if (i <= n_lim && i <= XAAC_SBR_MAX_LIMITERS) bad |= (unsigned)h->freq_band_tbl_lim[i] > 64u;
    if (i <= n_nf && i <= XAAC_SBR_MAX_NOISE_COEFFS) bad |= (unsigned)h->freq_band_tbl_noise[i] > 64u;
    if (i < n_if && i < XAAC_SBR_MAX_NOISE_VALUES) bad |= ((unsigned)h->bw_borders[i] > 64u) | ((unsigned)f->sbr_invf_mode[i] > 3u);
    if (i < n_pat && i < XAAC_SBR_MAX_PATCHES) {
      const xaac_sbr_patch *pp = &h->patch[i];
      bad |= ((unsigned)pp->src_start_band > 64u) | ((unsigned)pp->src_end_band > 64u) | ((unsigned)pp->guard_start_band > 64u) |
             ((unsigned)pp->dst_start_band > 64u) | ((unsigned)pp->dst_end_band > 64u) | ((unsigned)pp->num_bands_in_patch > 64u);
      /* a patch's high bands (low band + dst_end_band, the reference's name for the offset) stay inside the row */
      bad |= pp->src_end_band > pp->src_start_band && pp->src_end_band + pp->dst_end_band > 64;
    }
    if (i <= n_env && i <= XAAC_SBR_MAX_ENVELOPES) bad |= (unsigned)f->border_vec[i] > (ld ? (unsigned)h->num_time_slots : 19u); /* (low-delay: the matrix holds the frame's own rows) */
    if (i < n_env && i < XAAC_SBR_MAX_ENVELOPES) bad |= (unsigned)f->freq_res[i] > 1u;
    if (i <= n_nenv && i <= XAAC_SBR_MAX_NOISE_ENVELOPES) bad |= (unsigned)f->noise_border_vec[i] > 19u;
  }
  return cx.wave_or(bad) != 0;
}


/* sbr_dec.c:1050-1120, everything but the matrix: from the headrooms of the low bands' analysed slots (reserve) and overlap
   slots (reserve_ov1) to the two shifts the matrix takes.  Rescales the LPC history rows of the state and writes its
   ov_lb_scale / lb_scale. */
struct XsBfp {
  int sh_ov, sh_main; /* shift of slots 0..5 / 6..37 of bands [0, usb) (xs_adjust's meaning: left if positive) */
  int save_lb_scale, max_samp_val;
};
template <int HQ, class ST>
FX_HD XsBfp xs_bfp_shifts(const XsCx &cx, ST *st, int usb, int reserve, int reserve_ov1) {
  XsBfp b;
  b.max_samp_val = reserve < reserve_ov1 ? reserve : reserve_ov1;
  int32_t m = 1;
  XS_PAR(k, 0, usb) {
    m |= fx_abs_nrm(st->lpc_real[0][k]) | fx_abs_nrm(st->lpc_real[1][k]);
    if (HQ) m |= fx_abs_nrm(st->lpc_imag[0][k]) | fx_abs_nrm(st->lpc_imag[1][k]);
  }
  const int reserve_ov2 = xs_pnorm32(cx.wave_or(m));
  XS_T(27);
  if (reserve_ov2 < reserve_ov1) reserve_ov1 = reserve_ov2;
  const int lb_scale0 = cx.uni(st->lb_scale), ov_lb_scale0 = cx.uni(st->ov_lb_scale);
  const int shift1 = lb_scale0 + reserve, shift2 = ov_lb_scale0 + reserve_ov1;
  const int min_shift = shift1 < shift2 ? shift1 : shift2;
  const int shift_over = shift2 - min_shift;
  reserve -= (shift1 - min_shift);
  cx.sync();
  b.sh_ov = reserve_ov1 - shift_over;
  b.sh_main = reserve;
  {
    int sh = b.sh_ov;
    if (sh != 0) {
      if (sh > 31) sh = 31;
      if (sh < -31) sh = -31;
      XS_PAR(k, 0, usb)
        for (int i = 0; i < 2; i++) {
          st->lpc_real[i][k] = sh > 0 ? fx_shlw(st->lpc_real[i][k], sh) : (st->lpc_real[i][k] >> -sh);
          if (HQ) st->lpc_imag[i][k] = sh > 0 ? fx_shlw(st->lpc_imag[i][k], sh) : (st->lpc_imag[i][k] >> -sh);
        }
    }
  }
  b.save_lb_scale = (int16_t)(lb_scale0 + reserve);
  XS_ONE {
    st->ov_lb_scale = (int16_t)(ov_lb_scale0 + b.sh_ov);
    st->lb_scale = (int16_t)b.save_lb_scale;
  }
  return b;
}
/* xs_adjust (env_calc.c:1099) on one word; shift != 0 */
FX_HD int32_t xs_adjust_word(int32_t v, int shift) {
  if (shift > 31) shift = 31;
  if (shift < -31) shift = -31;
  return shift > 0 ? fx_shlw(v, shift) : (v >> -shift);
}

/* sbr_dec.c:1121-1245, :1283-1308: what follows the block floating point of the low bands -- HF generation, envelope
   adjustment, the state's scale factors, LPC history.  The matrix has been through xs_adjust with b's shifts and its
   analysed slots are zero from band 32 on. */
template <class ST, class Q>
FX_HD int xs_sbr_core_tail(const XsCx &cx, const xaac_sbr_header *h, const xaac_sbr_frame *f, const int16_t *env_sf_all,
                           const int16_t *noise_floor_all, ST *st, const Q &x, XsWork *w, const int16_t *rand_hi,
                           const XsBfp &b, XsPendingAdjust *pend = nullptr, const int32_t *sf_words = nullptr) {
  const int save_lb_scale = b.save_lb_scale, max_samp_val = b.max_samp_val;
  (void)max_samp_val;
  if (cx.uni(f->apply_processing)) {
    XsLv deg64;
    deg64.fill(0);
    const int n_env = cx.uni(f->num_env), nts = cx.uni(h->num_time_slots), ts = cx.uni(h->time_step);
    const int16_t last = fx_sat16((int32_t)cx.uni(f->border_vec[n_env]) - nts);
    if constexpr (!Q::HQ)
      xs_low_pow_hf_generator(cx, h, st, x, w, deg64, cx.uni(f->border_vec[0]) * ts, ts * last,
                              cx.uni(f->max_qmf_subband_aac), f->sbr_invf_mode, st->prev_invf_mode, max_samp_val);
    else
      xs_hf_generator_hq(cx, h, st, x, w, cx.uni(f->border_vec[0]) * ts, ts * last, cx.uni(f->max_qmf_subband_aac),
                         f->sbr_invf_mode, st->prev_invf_mode);
    XS_T(2);
    XS_ONE st->hb_scale = (int16_t)((st->ov_lb_scale < st->lb_scale ? st->ov_lb_scale : st->lb_scale) - 2);
    cx.sync();
    if (xs_calc_sbrenvelope(cx, h, f, env_sf_all, noise_floor_all, st, x, w, rand_hi, deg64, pend, sf_words)) return -1;
    XS_PAR(i, 0, h->num_if_bands) st->prev_invf_mode[i] = f->sbr_invf_mode[i];
    XS_ONE {
      st->prev_coupling_mode = f->coupling_mode;
      st->prev_max_qmf_subband_aac = f->max_qmf_subband_aac;
      st->prev_end_position = f->border_vec[f->num_env];
      st->prev_amp_res = f->amp_res;
    }
  } else {
    XS_ONE st->hb_scale = (int16_t)save_lb_scale;
  }
  cx.sync();
  xs_lpc_save(cx, st, x, cx.uni(st->codec_usb), Q::LD ? cx.uni(h->num_time_slots) * cx.uni(h->time_step) : 32);
  cx.sync();
  XS_T(15);
  return 0;
}

/* The part of ixheaacd_sbr_dec between the two QMF banks (sbr_dec.c:1050-1245), low-power (Q = XsQmf) or
   HQ (Q = XsQmfHq) mode.
   On entry x holds the 6 overlap slots (already through xs_rescale_x_overlap) and the 32 freshly
   analysed slots (bands 0..31); on exit x is ready for the synthesis bank and the state carries the
   new scale factors, LPC history and envelope-adjuster memory.  rand_hi[i] = xaac_sbr_rand_ph[i] >> 16
   (the only part of that table this mode uses; an LDS copy on the GPU).  Returns 0 or -1.
   (The GPU core kernel takes the headrooms and applies the shifts while the matrix words pass through its registers on their
   way into LDS, and calls xs_bfp_shifts / xs_sbr_core_tail itself: sbr_core_kernel.hip.) */
template <class ST, class Q>
FX_HD int xs_sbr_core(const XsCx &cx, const xaac_sbr_header *h, const xaac_sbr_frame *f, const int16_t *env_sf_all,
                      const int16_t *noise_floor_all, ST *st, const Q &x, XsWork *w, const int16_t *rand_hi,
                      int *save_lb_scale_out) {
  /* the frame grid this implementation (and the reference's MAX_ENV_COLS buffers, sbr_dec.c:733-748) is laid
     out for: 16 time slots of 2 QMF slots; anything else is refused like the reference refuses its own limits */
  const int no_bins = cx.uni(h->num_time_slots) * cx.uni(h->time_step);
  if (Q::LD ? ((no_bins != 16 && no_bins != 15) || cx.uni(h->time_step) != 1 || cx.uni(h->num_columns) != no_bins)
            : (no_bins != 32 || cx.uni(h->num_columns) != 32))
    return -1;
  /* xs_side_info_bad has been run by the caller, before xs_rescale_x_overlap touched the overlap slots */
  const int usb = cx.uni(st->codec_usb);
  constexpr int OV = Q::OV;
  const int reserve = xs_headroom(cx, x, 0, usb, OV, OV + no_bins);
  const int reserve_ov1 = Q::LD ? reserve : xs_headroom(cx, x, 0, usb, 0, OV); /* sbr_dec.c:1059-1080: one scan without overlap slots */
  const XsBfp b = xs_bfp_shifts<Q::HQ>(cx, st, usb, reserve, reserve_ov1);
  xs_adjust(cx, x, 0, usb, 0, OV, b.sh_ov);
  xs_adjust(cx, x, 0, usb, OV, OV + no_bins, b.sh_main);
  *save_lb_scale_out = b.save_lb_scale;
  xs_clear(cx, x, 32, Q::NB, OV, OV + no_bins); /* bands 32 and up of the analysed slots */
  cx.sync();
  XS_T(1);
  return xs_sbr_core_tail(cx, h, f, env_sf_all, noise_floor_all, st, x, w, rand_hi, b);
}

#endif /* XAAC_SBR_CORE_H */
