/*
 * tests/fuzz/tns_filter_check.cpp -- TEST INFRASTRUCTURE: the host parser's TNS all-pole filter and spectral code word
 * tables against the plain forms they replaced.
 *   - tns_ar_filter (libxaac_amd/host/aac_core.cpp) runs a line's multiply-adds without clamps while every state is inside
 *     a range in which no partial sum can saturate, and the reference's chain from the first state outside it.  Here it is
 *     compared, word for word, with the chain as ixheaacd_aac_tns.c:371-420 runs it (shift register and all), on random
 *     regions: quiet ones, loud ones that saturate, orders 1..12 (with the rounding up to a multiple of four), both directions.
 *   - every 10-bit window of every spectral book: the combined table's (length, values, sign count) against the general
 *     decode of the same bits.
 * Includes the translation unit to reach its file-local functions.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../libxaac_amd/host/aac_core.cpp"

namespace {
void tns_reference(int32_t *x, int size, int inc, int16_t *lpc, int order, int shift_value, int scale_spec) {
  int32_t state[32 + 4 + 1];
  if (order & 3) {
    int i;
    for (i = order + 1; i < (order & ~3) + 4; i++) lpc[i] = 0;
    if (i < 32) {
      lpc[i] = 0;
      order = (order & ~3) + 4;
    } else {
      order = 31;
    }
  }
  const int n = size > order ? size : order;
  for (int i = 0; i < n; i++) {
    int32_t y = fx_shl_sat(*x, scale_spec);
    int32_t acc = 0;
    for (int j = i < order ? i : order; j > 0; j--) {
      acc = fx_add_sat(acc, fx_mul32x16(state[j - 1], lpc[j]));
      state[j] = state[j - 1];
    }
    y = fx_sub_sat(y, fx_shl_sat(acc, 1));
    state[0] = fx_shl_sat(y, shift_value);
    *x = y >> scale_spec;
    x += inc;
  }
}
}  // namespace

int main(int argc, char **argv) {
  const int cases = argc > 1 ? atoi(argv[1]) : 20000;
  std::mt19937 rng(20250929);
  long loud = 0;
  for (int c = 0; c < cases; c++) {
    const int order = 1 + (int)(rng() % 12), size = 1 + (int)(rng() % (c % 7 == 0 ? 1000 : 96)), inc = (rng() & 1) ? 1 : -1;
    const int kind = (int)(rng() % 4); /* 0, 1: the usual headroom; 2: little; 3: none, large coefficients */
    const int shift_value = (int)(rng() % (kind == 3 ? 6 : 3)), scale_spec = kind >= 2 ? (int)(rng() % 4) : 4 + (int)(rng() % 8);
    int16_t lpc_a[40], lpc_b[40];
    for (int j = 0; j < 40; j++) lpc_a[j] = (int16_t)((int)(rng() % 65536) - 32768) / (kind == 3 ? 1 : (kind == 2 ? 2 : 6));
    memcpy(lpc_b, lpc_a, sizeof(lpc_a));
    static int32_t a[1200], b[1200];
    const int amp_bits = kind >= 2 ? 31 : 20 + (int)(rng() % 8);
    for (int i = 0; i < 1200; i++) a[i] = (int32_t)(rng() >> (32 - amp_bits)) - (int32_t)(1u << (amp_bits - 1));
    memcpy(b, a, sizeof(a));
    int32_t *pa = inc > 0 ? a + 40 : a + 40 + 1100, *pb = inc > 0 ? b + 40 : b + 40 + 1100;
    tns_reference(pa, size, inc, lpc_a, order, shift_value, scale_spec);
    tns_ar_filter(pb, size, inc, lpc_b, order, shift_value, scale_spec);
    if (memcmp(a, b, sizeof(a)) || memcmp(lpc_a, lpc_b, sizeof(lpc_a))) {
      fprintf(stderr, "case %d differs (order %d size %d inc %d kind %d)\n", c, order, size, inc, kind);
      return 1;
    }
    for (int i = 0; i < 1200; i++) loud += a[i] == FX_MAX32 || a[i] == FX_MIN32;
  }
  /* the combined tables */
  uint8_t pad[16];
  long entries = 0;
  for (int cb = 1; cb <= 11; cb++) {
    xh_core_init(nullptr, -1); /* (builds the books) */
    const Book &k = g_book[cb];
    const bool uns = !(cb <= 2 || cb == 5 || cb == 6);
    for (int w10 = 0; w10 < 1024; w10++) {
      const FastEntry f = k.fast[w10];
      if (!f.len) continue;
      entries++;
      for (int signs = 0; signs < 16; signs++) { /* the window, then four bits that would be the sign bits */
        memset(pad, 0, sizeof(pad));
        const uint32_t top = ((uint32_t)w10 << 22);
        /* place `signs` right behind the code word */
        const uint32_t word = (top & ~(0xffffffffu >> f.len)) | ((uint32_t)signs << (28 - f.len));
        pad[0] = (uint8_t)(word >> 24), pad[1] = (uint8_t)(word >> 16), pad[2] = (uint8_t)(word >> 8), pad[3] = (uint8_t)word;
        XhBits ba(pad, sizeof(pad)), bb(pad, sizeof(pad));
        int32_t xa[4] = {0, 0, 0, 0}, xb[4] = {0, 0, 0, 0};
        int err = 0;
        spectral_long_word(&ba, k, cb, xa, nullptr, &err);
        const int n = cb <= 4 ? 4 : 2;
        if (spectral_long(&bb, cb, n, xb, nullptr) || err || ba.pos != bb.pos || memcmp(xa, xb, sizeof(xa)) ||
            ba.pos != (size_t)f.len + (uns ? f.nsign : 0)) {
          fprintf(stderr, "book %d window %03x signs %x: table and general decode differ\n", cb, w10, signs);
          return 1;
        }
      }
    }
  }
  printf("%d filter cases equal to the chain (%ld saturated words among them), %ld table entries equal to the general decode\n", cases, loud,
         entries);
  return 0;
}
