/*
 * bits.h -- MSB-first bit reader of the host-side syntax parsers (ADTS / raw_data_block / SBR / PS payloads).
 * Reads past the end return zeros and latch `overrun`, which the callers turn into a frame error (the reference
 * tracks cnt_bits < 0 for the same purpose: decoder/ixheaacd_bitbuffer.c).
 */
#ifndef XAAC_HOST_BITS_H
#define XAAC_HOST_BITS_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>

struct XhBits {
  const uint8_t *p;
  size_t n_bits, pos;
  bool overrun;

  XhBits(const uint8_t *data, size_t bytes) : p(data), n_bits(bytes * 8), pos(0), overrun(false) {}

  /* the next 32 bits, first bit in the MSB, without consuming them */
  uint32_t peek32() const { return peek32_at(pos); }
  /* ... at any bit position (decoding loops that keep the position in a register and hand it back with `pos = ...`) */
  uint32_t peek32_at(size_t pos) const {
    const size_t byte = pos >> 3;
    const size_t total = n_bits >> 3;
    uint64_t w;
    if (byte + 8 <= total) { /* eight bytes at once, most significant first */
      memcpy(&w, p + byte, 8);
      w = __builtin_bswap64(w);
      return (uint32_t)((w << (pos & 7)) >> 32);
    }
    w = 0;
    for (size_t k = 0; k < 5; k++) w = (w << 8) | (byte + k < total ? p[byte + k] : 0);
    return (uint32_t)(w >> (8 - (pos & 7)));
  }
  void skip(size_t n) {
    pos += n;
    if (pos > n_bits) overrun = true;
  }
  uint32_t get(int n) { /* 0 <= n <= 32 */
    if (n == 0) return 0;
    const uint32_t v = peek32() >> (32 - n);
    skip((size_t)n);
    return v;
  }
  int get1() { return (int)get(1); }
  size_t left() const { return pos < n_bits ? n_bits - pos : 0; }
  void align() { skip((8 - (pos & 7)) & 7); }
};

#endif /* XAAC_HOST_BITS_H */
