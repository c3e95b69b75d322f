/*
 * fx.h -- fixed-point primitive semantics of the libxaac decoder back-end,
 * restated for three compilers at once: gcc (C, oracle), g++ (host) and hipcc
 * (gfx950 device code).
 *
 * Each primitive names the reference definition it must agree with bit-for-bit
 * (all paths relative to the reference tree):
 *   common/ixheaac_basic_ops32.h, common/ixheaac_basic_ops16.h,
 *   common/ixheaac_basic_ops40.h, common/ixheaac_basic_ops.h,
 *   decoder/ixheaacd_aac_imdct.c:80-106 (file-local 32x16 "l" forms).
 *
 * The reference is built with -fwrapv and leans on wrapping int32 +,-,<< in its
 * FFTs.  Here every wrapping op is done on uint32_t so the
 * behaviour does not depend on compiler flags, on host or device.
 */
#ifndef XAAC_FX_H
#define XAAC_FX_H

#include <stdint.h>

#if defined(__HIPCC__)
#define FX_HD __host__ __device__ __forceinline__
#define FX_MEMBER __host__ __device__ __forceinline__
#else
#define FX_HD static inline
#define FX_MEMBER inline
#endif

#define FX_MAX32 ((int32_t)0x7fffffff)
#define FX_MIN32 ((int32_t)0x80000000)

/* ---- wrapping ring ops (raw + - << under -fwrapv) ---------------------- */
FX_HD int32_t fx_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
FX_HD int32_t fx_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
FX_HD int32_t fx_neg(int32_t a) { return (int32_t)(0u - (uint32_t)a); }
/* x << n for 0 <= n <= 31, wrapping */
FX_HD int32_t fx_shlw(int32_t a, int n) { return (int32_t)((uint32_t)a << n); }

/* ---- saturating add/sub: basic_ops32.h:197-203, :225-231 ---------------- */
FX_HD int32_t fx_sat64(int64_t v) {
  if (v > (int64_t)FX_MAX32) return FX_MAX32;
  if (v < (int64_t)FX_MIN32) return FX_MIN32;
  return (int32_t)v;
}
FX_HD int32_t fx_add_sat(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_elementwise_add_sat(a, b); /* v_add_i32 ... clamp: the same two-sided clamp */
#else
  return fx_sat64((int64_t)a + (int64_t)b);
#endif
}
FX_HD int32_t fx_sub_sat(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_elementwise_sub_sat(a, b); /* v_sub_i32 ... clamp */
#else
  return fx_sat64((int64_t)a - (int64_t)b);
#endif
}
/* basic_ops32.h:317-327 */
FX_HD int32_t fx_neg_sat(int32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_elementwise_sub_sat(0, a);
#else
  return a == FX_MIN32 ? FX_MAX32 : -a;
#endif
}
/* basic_ops32.h:295-307 */
FX_HD int32_t fx_abs_sat(int32_t a) { return a == FX_MIN32 ? FX_MAX32 : (a < 0 ? -a : a); }
/* basic_ops32.h:283-293 */
FX_HD int32_t fx_abs_nrm(int32_t a) { return a < 0 ? ~a : a; }

/* ---- shifts ------------------------------------------------------------ */
/* basic_ops32.h:39-49: count taken mod 256, >31 gives 0, wrapping otherwise */
FX_HD int32_t fx_shl(int32_t a, int b) {
  b &= 0xff;
  return b > 31 ? 0 : fx_shlw(a, b);
}
/* basic_ops32.h:51-65: arithmetic, count mod 256, >=31 gives the sign */
FX_HD int32_t fx_shr(int32_t a, int b) {
  b &= 0xff; /* counts of 31 and more leave the sign, which is what a shift by 31 leaves */
  return a >> (b < 31 ? b : 31);
}
/* basic_ops32.h:67-78 (0 <= b <= 31 on every path that reaches it) */
FX_HD int32_t fx_shl_sat(int32_t a, int b) {
  if (a > (FX_MAX32 >> b)) return FX_MAX32;
  if (a < (FX_MIN32 >> b)) return FX_MIN32;
  return fx_shlw(a, b);
}
/* basic_ops32.h:377-394: ROUNDING right shift despite the name */
FX_HD int32_t fx_shr_rnd(int32_t a, int b) {
  b &= 0xff;
  if (b >= 31) return a < 0 ? -1 : 0;
  if (b <= 0) return a;
  return fx_add_sat(a, (int32_t)1 << (b - 1)) >> b;
}
/* basic_ops.h:114-126  shl32_dir_sat_limit */
FX_HD int32_t fx_shl_dir_sat_limit(int32_t a, int b) {
  if (b < 0) {
    b = -b;
    if (b > 31) b = 31;
    return fx_shr(a, b);
  }
  return fx_shl_sat(a, b);
}
/* basic_ops32.h:80-90 / :92-102 */
FX_HD int32_t fx_shl_dir(int32_t a, int b) { return b < 0 ? fx_shr(a, -b) : fx_shl(a, b); }
FX_HD int32_t fx_shr_dir(int32_t a, int b) { return b < 0 ? fx_shl(a, -b) : fx_shr(a, b); }

/* (WORD32)v of a float as the reference's x86-64 build computes it (cvttss2si): truncation, and the "integer
   indefinite" 0x80000000 for NaN and for everything outside int32 -- where C leaves the conversion undefined and the
   GPU's own conversion would saturate instead */
FX_HD int32_t fx_f2i_trunc(float v) { return (v >= 2147483648.0f || v < -2147483648.0f || v != v) ? FX_MIN32 : (int32_t)v; }

/* ---- norm -------------------------------------------------------------- */
/* basic_ops32.h:236-255: redundant sign bits; 0 and -1 give 31 */
FX_HD int fx_norm32(int32_t a) {
  uint32_t u = (uint32_t)(a < 0 ? ~a : a);
  if (u == 0) return 31;
#if defined(__HIP_DEVICE_COMPILE__)
  return __clz((int)u) - 1;
#else
  return __builtin_clz(u) - 1;
#endif
}

/* ---- 16-bit helpers: basic_ops16.h ------------------------------------- */
/* :231-235 */
FX_HD int16_t fx_round16(int32_t a) { return (int16_t)(fx_add_sat(a, 0x8000) >> 16); }
/* :206-216 */
FX_HD int16_t fx_neg16(int16_t a) { return a == (int16_t)-32768 ? (int16_t)32767 : (int16_t)-a; }
FX_HD int16_t fx_sat16(int32_t a) { return a > 32767 ? (int16_t)32767 : (a < -32768 ? (int16_t)-32768 : (int16_t)a); }

/* ---- multiplies -------------------------------------------------------- */
FX_HD int32_t fx_mulhi(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __mulhi(a, b);
#else
  return (int32_t)(((int64_t)a * (int64_t)b) >> 32);
#endif
}
/* basic_ops40.h:34-40  (a*b)>>16, b a signed 16-bit value.
   == hi32(a * (b<<16)), one v_mul_hi_i32 on the device. */
FX_HD int32_t fx_mul32x16(int32_t a, int16_t b) { return fx_mulhi(a, (int32_t)((uint32_t)(uint16_t)b << 16)); }
/* same with the coefficient already parked in the top half of a word
   (low half ignored): covers mult32x16hin32 (basic_ops32.h:134-140). */
FX_HD int32_t fx_mul32xhi(int32_t a, int32_t packed) { return fx_mulhi(a, (int32_t)((uint32_t)packed & 0xffff0000u)); }
/* coefficient in the low half: aac_imdct.c:80-86 mult32x16lin32 */
FX_HD int32_t fx_mul32xlo(int32_t a, int32_t packed) { return fx_mulhi(a, (int32_t)((uint32_t)packed << 16)); }
/* aac_imdct.c:95-106 / basic_ops32.h:142-157: full product, NO shift, clamped */
FX_HD int32_t fx_mul32x16_nosh_sat(int32_t a, int16_t b) { return fx_sat64((int64_t)a * (int64_t)b); }
/* basic_ops40.h:78-84 / :68-74 */
FX_HD int32_t fx_mul32(int32_t a, int32_t b) { return fx_mulhi(a, b); }
FX_HD int32_t fx_mul32_shl(int32_t a, int32_t b) { return fx_shlw(fx_mulhi(a, b), 1); }
/* basic_ops40.h:23-29 */
FX_HD int32_t fx_mul32x16_shl(int32_t a, int16_t b) { return fx_shlw(fx_mul32x16(a, b), 1); }

#endif /* XAAC_FX_H */
