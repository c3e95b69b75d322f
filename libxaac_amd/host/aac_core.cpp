/*
 * aac_core.cpp -- AAC-LC raw_data_block -> spectra at the IMDCT seam, on the host (see aac_core.h).
 * Restates decoder/ixheaacd_aacdecoder.c:362-647 (element loop), ixheaacd_channel.c (ics_info :356, pulse :183,
 * block data :214, spectral data :749, TNS data :961, inverse quantiser :1055, channel pair tools :602),
 * ixheaacd_longblock.c (sections :62, scale factors :155), ixheaacd_block.c (Huffman + inverse quantisation :129-1130,
 * scale factor gains :1242), ixheaacd_stereo.c (M/S :54, intensity :129), ixheaacd_pns_js_thumb.c (PNS :74-200,
 * TNS :202-514), ixheaacd_aac_tns.c (:147 parcor -> LPC, :371 filter, :422 headroom), ixheaacd_aacpluscheck.c:59 (FIL).
 * Tables: tables_aac.inc (generated from the compiled reference's ROM).
 */
#include "aac_core.h"

#include <stddef.h>
#include <string.h>

#include "../csrc/fx.h"
#include "tables_aac.inc"

namespace {

/* ---- code books --------------------------------------------------------------------------------------------------- */
/* what the first ten bits of a spectral code word decide, if they hold all of it: its length, the two or four quantised
   values (signed books: with their sign; unsigned books: magnitudes, `nsign` sign bits follow the code word, one per
   non-zero value in line order).  len 0: a longer code word, or a pair of book 11 with an escape -- the general path. */
struct FastEntry {
  int8_t v[4];
  uint8_t len, nsign;
};
struct Book {
  const uint32_t *code;
  const uint8_t *len;
  const uint16_t *idx;
  int n;
  uint16_t lut[1024]; /* the first ten bits -> entry (length <= 10), or 0xffff */
  FastEntry fast[1024];
};
Book g_book[12];
int32_t g_deq[33]; /* deq_long(q, 0) for q = -16 .. 16 at [q + 16] */

#define XH_BOOK(k) {xh_hcb##k##_code, xh_hcb##k##_len, xh_hcb##k##_idx, (int)(sizeof(xh_hcb##k##_len)), {0}, {{{0}, 0, 0}}}

/* the values of value-order index idx of spectral book cb (block.c:129-1130: quads base 3, pairs base 9 / 8 / 13 / 17) */
inline int unpack_values(int cb, int idx, int *v) {
  if (cb <= 4) {
    v[0] = idx / 27, idx -= v[0] * 27;
    v[1] = idx / 9, idx -= v[1] * 9;
    v[2] = idx / 3, v[3] = idx - v[2] * 3;
    if (cb <= 2)
      for (int j = 0; j < 4; j++) v[j] -= 1;
    return 4;
  }
  const int mod = cb <= 6 ? 9 : (cb <= 8 ? 8 : (cb <= 10 ? 13 : 17));
  v[0] = idx / mod, v[1] = idx % mod;
  if (cb <= 6) v[0] -= 4, v[1] -= 4;
  return 2;
}

void build_books() {
  static const Book init[12] = {XH_BOOK(0), XH_BOOK(1), XH_BOOK(2), XH_BOOK(3), XH_BOOK(4),  XH_BOOK(5),
                                XH_BOOK(6), XH_BOOK(7), XH_BOOK(8), XH_BOOK(9), XH_BOOK(10), XH_BOOK(11)};
  for (int b = 0; b < 12; b++) {
    g_book[b] = init[b];
    Book &k = g_book[b];
    for (int i = 0; i < 1024; i++) k.lut[i] = 0xffff;
    for (int e = 0; e < k.n; e++)
      if (k.len[e] <= 10) {
        const uint32_t first = k.code[e] >> 22, count = 1u << (10 - k.len[e]);
        for (uint32_t j = 0; j < count; j++) k.lut[first + j] = (uint16_t)e;
      }
    if (b == 0) continue; /* the scale factor book has no value tuples */
    const bool is_signed = b == 1 || b == 2 || b == 5 || b == 6;
    for (int i = 0; i < 1024; i++) {
      FastEntry &f = k.fast[i];
      f = FastEntry{{0, 0, 0, 0}, 0, 0};
      const int e = k.lut[i];
      if (e == 0xffff) continue;
      int v[4] = {0, 0, 0, 0};
      const int n = unpack_values(b, k.idx[e], v);
      int nz = 0, esc = 0;
      for (int j = 0; j < n; j++) nz += v[j] != 0, esc += (b == 11 && v[j] == 16);
      if (esc) continue;
      for (int j = 0; j < n; j++) f.v[j] = (int8_t)v[j];
      f.len = k.len[e];
      f.nsign = is_signed ? 0 : (uint8_t)nz;
    }
  }
  for (int q = -16; q <= 16; q++) g_deq[q + 16] = q <= 0 ? -xh_pow43_q13[-q] : xh_pow43_q13[q];
}

/* one code word at the reader's position: its index in the book's value order */
inline int huff(const Book &k, XhBits *br) {
  const uint32_t w = br->peek32();
  int e = k.lut[w >> 22];
  if (e == 0xffff) { /* the code words are sorted: the last one not above the window is the prefix */
    int lo = 0, hi = k.n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (k.code[mid] <= w) lo = mid;
      else hi = mid - 1;
    }
    e = lo;
  }
  br->skip(k.len[e]);
  return k.idx[e];
}

/* ---- inverse quantiser ---------------------------------------------------------------------------------------------- */
/* |q|^(4/3) in Q13 of a magnitude: the table up to 128, beyond it the reference's interpolation between table entries
   of q / 8 (q < 1024) or q / 64 (channel.c:1055-1093); *err is set past 8191 + 32 */
inline int32_t pow43(int32_t q, int *err) {
  if (q <= 128) return xh_pow43_q13[q];
  if (q > 8191 + 32) {
    *err = 1;
    return q; /* the reference leaves the value as it is and raises the error (channel.c:1066) */
  }
  const int shift = q < 1024 ? 3 : 6;
  const int32_t q1 = q >> shift;
  const int16_t interp = (int16_t)(q - (q1 << shift));
  /* magnitudes 8192 .. 8223 make q1 + 1 = 129: the reference reads the word behind its 129-entry table, which is the first
     scale-factor gain of the same ROM struct (ixheaacd_aac_rom.h:26-27) */
  const int32_t upper = q1 + 1 <= 128 ? xh_pow43_q13[q1 + 1] : xh_scale_tab[0];
  int32_t t = (upper - xh_pow43_q13[q1]) * (int32_t)interp;
  t = fx_add(t, fx_shlw(xh_pow43_q13[q1], shift));
  return fx_shlw(t, shift == 3 ? 1 : 2);
}

/* the escape sequence behind a code word of book 11 (block.c:190-206): N ones, a zero, N + 4 bits */
inline int32_t escape(XhBits *br) {
  int n = 0;
  while (n < 9 && br->get1()) n++; /* the reference looks at 9 bits at most */
  if (n == 9) { /* nine ones: it consumes ten bits and takes 13 more (norm32 of the all-ones window) */
    br->skip(1);
  }
  const int bits = n + 4;
  return (int32_t)br->get(bits) + ((int32_t)1 << bits);
}

/* a decoded value q (sign applied) with the pulse amplitude t of its line -> the dequantised line of a long block
   (block.c:404-417, :674-686, :728-742, :932-978): zero and negative values come out negated */
inline int32_t deq_long(int32_t q, int t) { return q <= 0 ? -xh_pow43_q13[t - q] : xh_pow43_q13[q + t]; }

/* one code word of a long block's section the general way: any length, escapes (block.c:129-1130) */
inline void spectral_long_word(XhBits *br, const Book &k, int cb, int32_t *x, const uint8_t *pulse, int *err) {
  static const uint8_t no_pulse[4] = {0, 0, 0, 0};
  if (!pulse) pulse = no_pulse;
  const int idx = huff(k, br);
  int v[4];
  const int n = unpack_values(cb, idx, v);
  if (cb <= 10) {
    const bool is_signed = cb <= 2 || cb == 5 || cb == 6;
    for (int j = 0; j < n; j++) {
      int q = v[j];
      if (!is_signed && q && br->get1()) q = -q;
      x[j] = deq_long(q, pulse[j]);
    }
    return;
  }
  int neg[2] = {0, 0};
  for (int j = 0; j < 2; j++)
    if (v[j]) neg[j] = br->get1();
  for (int j = 0; j < 2; j++) {
    if (v[j] == 16) {
      const int32_t m = escape(br) + pulse[j];
      const int32_t p = pow43(m, err);
      x[j] = neg[j] ? fx_neg(p) : p;
    } else {
      x[j] = deq_long(neg[j] ? -v[j] : v[j], pulse[j]);
    }
  }
}

/* the lines of one section of a long block (ixheaacd_huffman_dec_word2): width lines at x, pulse amplitudes at pulse (or
   none).  N values per code word; UNS: an unsigned book, sign bits behind the code word.  Code words of up to ten bits take
   the book's combined table -- length, values and the number of sign bits from one look-up, the signs from the same
   32-bit window -- with the bit position in a register; the rest goes the general way. */
template <int N, bool UNS>
int spectral_long_section(XhBits *br, int cb, int width, int32_t *x, const uint8_t *pulse) {
  const Book &k = g_book[cb];
  int err = 0;
  /* a 64-bit reservoir, the next bit on top: refilled four bytes at a time while the frame has eight left from the fill
     position on, else (the last bytes of a frame) this section goes through the reader's own window */
  const uint8_t *const base = br->p;
  const size_t total = br->n_bits >> 3;
  size_t pos = br->pos, fill = (pos >> 3) + 8; /* `fill`: the first byte not in the reservoir */
  uint64_t buf = 0;
  int have = 0; /* valid bits in buf */
  bool fast = (pos >> 3) + 8 <= total;
  if (fast) {
    uint64_t w;
    memcpy(&w, base + (pos >> 3), 8);
    buf = __builtin_bswap64(w) << (pos & 7);
    have = 64 - (int)(pos & 7);
  }
  for (int i = 0; i < width; i += N) {
    uint32_t w;
    if (fast) {
      if (have < 32) {
        if (fill + 4 <= total) {
          uint32_t t;
          memcpy(&t, base + fill, 4);
          buf |= (uint64_t)__builtin_bswap32(t) << (32 - have);
          have += 32, fill += 4;
        } else {
          fast = false;
        }
      }
    }
    w = fast ? (uint32_t)(buf >> 32) : br->peek32_at(pos);
    const FastEntry f = k.fast[w >> 22];
    if (f.len) {
      uint32_t sb = w << f.len; /* the sign bits, the first one on top */
      const int used = f.len + f.nsign;
      pos += (size_t)used;
      buf <<= used, have -= used;
      for (int j = 0; j < N; j++) {
        int q = f.v[j];
        if (UNS) {
          const uint32_t nz = q != 0;
          q = ((sb >> 31) & nz) ? -q : q;
          sb <<= nz;
        }
        x[i + j] = pulse ? deq_long(q, pulse[i + j]) : g_deq[q + 16];
      }
    } else {
      br->pos = pos;
      spectral_long_word(br, k, cb, x + i, pulse ? pulse + i : nullptr, &err);
      pos = br->pos;
      fast = (pos >> 3) + 8 <= total; /* the reservoir starts over behind the long code word */
      if (fast) {
        uint64_t t;
        memcpy(&t, base + (pos >> 3), 8);
        buf = __builtin_bswap64(t) << (pos & 7);
        have = 64 - (int)(pos & 7);
        fill = (pos >> 3) + 8;
      }
    }
  }
  br->pos = pos;
  if (pos > br->n_bits) br->overrun = true;
  return err ? XH_ERR_ESCAPE : 0;
}

int spectral_long(XhBits *br, int cb, int width, int32_t *x, const uint8_t *pulse) {
  if (cb <= 2) return spectral_long_section<4, false>(br, cb, width, x, pulse);
  if (cb <= 4) return spectral_long_section<4, true>(br, cb, width, x, pulse);
  if (cb <= 6) return spectral_long_section<2, false>(br, cb, width, x, pulse);
  return spectral_long_section<2, true>(br, cb, width, x, pulse);
}

/* the lines of bands start .. stop-1 of one window group of a short frame (ixheaacd_decode_huffman): per band, per
   window of the group, the band's lines; x is the group's first window */
int spectral_short(XhBits *br, int cb, int32_t *x, const int16_t *swb, int start, int stop, int group_len) {
  const Book &k = g_book[cb];
  int err = 0;
  for (int sfb = start; sfb < stop; sfb++) {
    const int lo = swb[sfb], width = swb[sfb + 1] - lo;
    for (int w = 0; w < group_len; w++) {
      int32_t *d = x + 128 * w + lo;
      if (cb <= 4) {
        for (int i = 0; i < width; i += 4) {
          int idx = huff(k, br);
          int v[4];
          v[0] = idx / 27, idx -= v[0] * 27;
          v[1] = idx / 9, idx -= v[1] * 9;
          v[2] = idx / 3, v[3] = idx - v[2] * 3;
          for (int j = 0; j < 4; j++) {
            int q = v[j];
            if (cb <= 2) q -= 1;
            else if (q && br->get1()) q = -q;
            d[i + j] = q < 0 ? -xh_pow43_q13[-q] : xh_pow43_q13[q];
          }
        }
      } else if (cb <= 10) {
        const int mod = cb <= 6 ? 9 : (cb <= 8 ? 8 : 13);
        for (int i = 0; i < width; i += 2) {
          const int idx = huff(k, br);
          int v[2] = {idx / mod, idx % mod};
          for (int j = 0; j < 2; j++) {
            int q = v[j];
            if (cb <= 6) q -= 4;
            else if (q && br->get1()) q = -q;
            d[i + j] = q < 0 ? -xh_pow43_q13[-q] : xh_pow43_q13[q];
          }
        }
      } else {
        for (int i = 0; i < width; i += 2) {
          const int idx = huff(k, br);
          int v[2] = {idx / 17, idx % 17};
          int neg[2] = {0, 0};
          for (int j = 0; j < 2; j++)
            if (v[j]) neg[j] = br->get1();
          for (int j = 0; j < 2; j++) {
            const int32_t p = v[j] == 16 ? pow43(escape(br), &err) : xh_pow43_q13[v[j]];
            d[i + j] = neg[j] ? fx_neg(p) : p;
          }
        }
      }
    }
  }
  return err ? XH_ERR_ESCAPE : 0;
}

/* ---- scale factor gains: block.c:1242-1343 (at most two channels: q_factor 37) ------------------------------------ */
void apply_scale_factor(int sf, int32_t *x, int width) {
  if (sf < 24) {
    for (int j = 0; j < width; j++) x[j] = 0;
    return;
  }
  const int shift = 37 - (sf >> 2);
  const int16_t g = (int16_t)xh_scale_tab[sf & 3];
  if (shift > 0) {
    for (int j = 0; j < width; j++) x[j] = fx_shr(fx_mul32x16_shl(x[j], g), shift);
  } else if (shift < 0) {
    for (int j = 0; j < width; j++) x[j] = fx_shl(fx_mul32x16_shl(fx_shl(x[j], -shift - 1), g), 1);
  } else {
    for (int j = 0; j < width; j++) x[j] = fx_mul32x16_shl(x[j], g);
  }
}

/* ---- syntax --------------------------------------------------------------------------------------------------------- */
int read_ics(const XhCoreState *st, XhBits *br, XhIcs *ics) { /* channel.c:356-478 (AAC-LC branch) */
  const uint32_t v = br->get(4); /* ics_reserved_bit, window_sequence (2), window_shape */
  ics->window_sequence = (int)((v & 6) >> 1);
  ics->window_shape = (int)(v & 1);
  if (ics->window_sequence != XH_EIGHT_SHORT) {
    ics->num_swb = st->num_swb_long;
    ics->num_groups = 1;
    ics->group_len[0] = 1;
    const uint32_t w = br->get(7);
    ics->max_sfb = (int)((w & 0x7e) >> 1);
    if (w & 1) return XH_ERR_UNSUPPORTED; /* predictor_data_present: not an AAC-LC tool */
  } else {
    ics->num_swb = st->num_swb_short;
    const uint32_t w = br->get(11);
    ics->max_sfb = (int)((w & 0x780) >> 7);
    int groups = 0;
    for (int i = 0; i < 8; i++) ics->group_len[i] = 1;
    for (int i = 0, mask = 0x40; i < 7; i++, mask >>= 1) {
      if (w & mask) ics->group_len[groups]++;
      else groups++;
    }
    ics->num_groups = groups + 1;
  }
  if (ics->max_sfb > ics->num_swb) return XH_ERR_SYNTAX;
  return br->overrun ? XH_ERR_BITS : 0;
}

int read_sections(XhBits *br, XhChannel *c) { /* longblock.c:62-153 */
  const XhIcs &ics = c->ics;
  const int bits = ics.window_sequence == XH_EIGHT_SHORT ? 3 : 5, esc = (1 << bits) - 1;
  for (int g = 0; g < ics.num_groups; g++) {
    uint8_t *cb = c->cb + 16 * g;
    int sfb = 0;
    while (sfb < ics.max_sfb) {
      const int sect_cb = (int)br->get(4);
      int len = 0, incr = (int)br->get(bits);
      while (incr == esc) {
        len += esc;
        incr = (int)br->get(bits);
        if (br->overrun) return XH_ERR_BITS;
      }
      len += incr;
      sfb += len;
      if (br->overrun) return XH_ERR_BITS; /* zeros behind the end of the frame would make sections of length 0 for ever */
      if (sfb > ics.max_sfb) return XH_ERR_SYNTAX;
      if (sect_cb == XH_ESC_HCB + 1) return XH_ERR_SYNTAX;
      for (int i = sfb - len; i < sfb; i++) cb[i] = (uint8_t)sect_cb;
    }
  }
  return br->overrun ? XH_ERR_BITS : 0;
}

void read_scale_factors(XhBits *br, XhChannel *c) { /* longblock.c:155-306 (AAC-LC branch) */
  const XhIcs &ics = c->ics;
  int factor = c->global_gain, position = 0;
  for (int g = 0; g < ics.num_groups; g++) {
    for (int sfb = 0; sfb < ics.max_sfb; sfb++) {
      const int cb = c->cb[16 * g + sfb];
      int16_t *sf = &c->sf[16 * g + sfb];
      if (cb == XH_ZERO_HCB) {
        *sf = 0;
        continue;
      }
      int norm;
      if (cb == XH_NOISE_HCB && !c->pns_active) {
        norm = (int)br->get(9) - 256;
        c->pns_active = 1;
        c->noise_energy = (int16_t)(c->global_gain - 90); /* NOISE_OFFSET */
      } else {
        norm = huff(g_book[0], br) - 60;
      }
      if (cb > XH_NOISE_HCB) {
        position = (int16_t)(position + norm);
        *sf = (int16_t)-position;
      } else if (cb < XH_NOISE_HCB) {
        factor = (int16_t)(factor + norm);
        *sf = (int16_t)factor;
      } else {
        c->noise_energy = fx_sat16((int32_t)c->noise_energy + norm);
        *sf = c->noise_energy;
        c->pns_used[16 * g + sfb] = 1;
      }
    }
  }
}

int read_pulse(const XhCoreState *st, XhBits *br, XhPulse *p) { /* channel.c:183-212 */
  const uint32_t v = br->get(8);
  p->number = (int)(v >> 6);
  p->start_band = (int)(v & 0x3f);
  if (p->start_band >= 52) return XH_ERR_SYNTAX;
  int total = st->swb_long[p->start_band], err = 0;
  for (int i = 0; i <= p->number; i++) {
    const uint32_t w = br->get(9);
    p->offset[i] = (uint8_t)(w >> 4);
    p->amp[i] = (uint8_t)(w & 15);
    total += p->offset[i];
    if (total >= 1024) err = XH_ERR_SYNTAX;
  }
  return err;
}

int read_tns(XhBits *br, XhChannel *c) { /* channel.c:961-1053 */
  const bool is_short = c->ics.window_sequence == XH_EIGHT_SHORT;
  const int n_filt_bits = is_short ? 1 : 2, band_bits = is_short ? 4 : 6, order_bits = is_short ? 3 : 5;
  const int windows = is_short ? 8 : 1;
  for (int w = 0; w < windows; w++) {
    const int n_filt = c->tns.n_filt[w] = (int)br->get(n_filt_bits);
    if (!n_filt) continue;
    const int coef_res = br->get1();
    int top = c->ics.num_swb;
    for (int f = 0; f < n_filt; f++) {
      XhTnsFilter *flt = &c->tns.f[w][f];
      const int length = (int)br->get(band_bits);
      if (top < length) top = length;
      flt->start_band = top - length;
      flt->stop_band = top;
      top = flt->start_band;
      const int order = flt->order = (int)br->get(order_bits);
      if (order > 12) return XH_ERR_SYNTAX; /* MAX_ORDER_LONG, for short windows too (channel.c:1021) */
      if (order) {
        flt->direction = br->get1() ? -1 : 1;
        const int compress = br->get1();
        flt->resolution = coef_res;
        const int res = coef_res + 3 - compress;
        for (int i = 0; i < order; i++) {
          const int32_t v = (int32_t)br->get(res);
          flt->coef[i] = (int8_t)(fx_shlw(v, 32 - res) >> (32 - res)); /* sign extension of the res-bit field */
        }
      }
    }
  }
  return br->overrun ? XH_ERR_BITS : 0;
}

int read_spectrum(const XhCoreState *st, XhBits *br, XhChannel *c) { /* channel.c:749-905 */
  const XhIcs &ics = c->ics;
  int32_t *spec = c->spec();
  memset(c->spec_mem, 0, XH_SPEC_WORDS * sizeof(int32_t));
  if (ics.window_sequence != XH_EIGHT_SHORT) {
    uint8_t pulse[1024];
    if (c->pulse.present) { /* channel.c:727-747 */
      memset(pulse, 0, sizeof(pulse));
      int k = st->swb_long[c->pulse.start_band];
      for (int i = 0; i <= c->pulse.number; i++) {
        k += c->pulse.offset[i];
        pulse[k] = c->pulse.amp[i];
      }
    }
    for (int sfb = 0; sfb < ics.max_sfb;) {
      const int cb = c->cb[sfb], start = sfb;
      while (sfb < ics.max_sfb && c->cb[sfb] == cb) sfb++;
      const int lo = st->swb_long[start], width = st->swb_long[sfb] - lo;
      if (cb > XH_ZERO_HCB && cb < XH_NOISE_HCB) {
        const int e = spectral_long(br, cb, width, spec + lo, c->pulse.present ? pulse + lo : nullptr);
        if (e) return e;
      } else if (c->pulse.present) { /* a pulse on a line without spectral data: block.c:115-127, negated */
        for (int i = 0; i < width; i++) spec[lo + i] = -xh_pow43_q13[pulse[lo + i]];
      }
    }
    /* (bands without spectral data -- code book 0, noise, intensity -- hold zeros, and zeros they stay whatever the gain) */
    for (int sfb = 0, lo = 0; sfb < ics.max_sfb; lo += st->width_long[sfb], sfb++)
      if ((c->cb[sfb] > XH_ZERO_HCB && c->cb[sfb] < XH_NOISE_HCB) || c->pulse.present)
        apply_scale_factor(c->sf[sfb], spec + lo, st->width_long[sfb]);
  } else {
    int win = 0;
    for (int g = 0; g < ics.num_groups; g++) {
      const uint8_t *cbs = c->cb + 16 * g;
      for (int sfb = 0; sfb < ics.max_sfb;) {
        const int cb = cbs[sfb], start = sfb;
        while (sfb < ics.max_sfb && cbs[sfb] == cb) sfb++;
        if (cb > XH_ZERO_HCB && cb < XH_NOISE_HCB) {
          const int e = spectral_short(br, cb, spec + 128 * win, st->swb_short, start, sfb, ics.group_len[g]);
          if (e) return e;
        }
      }
      for (int w = 0; w < ics.group_len[g]; w++, win++)
        for (int sfb = 0, lo = 0; sfb < ics.max_sfb; lo += st->width_short[sfb], sfb++)
          apply_scale_factor(c->sf[16 * g + sfb], spec + 128 * win + lo, st->width_short[sfb]);
    }
  }
  return br->overrun ? XH_ERR_BITS : 0;
}

int read_channel_stream(const XhCoreState *st, XhBits *br, XhElement *el, int ch) { /* channel.c:481-563, :214-323 */
  XhChannel *c = &el->ch[ch];
  c->global_gain = (int)br->get(8);
  if (!el->common_window) {
    const int e = read_ics(st, br, &c->ics);
    if (e) return e;
  }
  if (c->ics.window_sequence == XH_EIGHT_SHORT) memset(c->sf, 0, sizeof(c->sf));
  int e = read_sections(br, c);
  if (e) return e;
  read_scale_factors(br, c);
  c->pulse.present = br->get1();
  if (c->pulse.present) {
    e = read_pulse(st, br, &c->pulse);
    if (e) return e;
  }
  c->tns.present = br->get1();
  if (c->tns.present) {
    e = read_tns(br, c);
    if (e) return e;
  }
  if (br->get1()) return XH_ERR_UNSUPPORTED; /* gain_control_data_present */
  if (br->overrun) return XH_ERR_BITS;
  return read_spectrum(st, br, c);
}

/* ---- tools ---------------------------------------------------------------------------------------------------------- */
void ms_stereo(const XhCoreState *st, XhElement *el) { /* stereo.c:54-116 */
  const XhIcs &ics = el->ch[0].ics;
  const int8_t *width = el->ch[1].ics.window_sequence == XH_EIGHT_SHORT ? st->width_short : st->width_long;
  int32_t *l = el->ch[0].spec(), *r = el->ch[1].spec();
  for (int g = 0; g < ics.num_groups; g++)
    for (int w = 0; w < ics.group_len[g]; w++) {
      int off = 0;
      for (int sfb = 0; sfb < ics.max_sfb; sfb++) {
        if (el->ms_used[g][sfb])
          for (int k = 0; k < width[sfb]; k++) {
            const int32_t a = l[off + k], b = r[off + k];
            l[off + k] = fx_add_sat(a, b);
            r[off + k] = fx_sub_sat(a, b);
          }
        off += width[sfb];
      }
      l += 128, r += 128;
    }
}

void intensity_stereo(const XhCoreState *st, XhElement *el) { /* stereo.c:129-243 */
  const XhChannel &rc = el->ch[1];
  const XhIcs &ics = rc.ics;
  const int8_t *width = ics.window_sequence == XH_EIGHT_SHORT ? st->width_short : st->width_long;
  int32_t *l = el->ch[0].spec(), *r = el->ch[1].spec();
  for (int g = 0; g < ics.num_groups; g++)
    for (int w = 0; w < ics.group_len[g]; w++) {
      int off = 0;
      for (int sfb = 0; sfb < ics.max_sfb; sfb++) {
        const int cb = rc.cb[16 * g + sfb];
        if (cb >= XH_INTENSITY_HCB2) {
          const int sf = rc.sf[16 * g + sfb];
          int32_t scale = xh_scale_tab[sf & 3];
          if (!(el->ms_used[g][sfb] ^ (cb & 1))) scale = fx_neg_sat(scale);
          const int scf_exp = -((sf >> 2) + 2);
          for (int k = 0; k < width[sfb]; k++) {
            int32_t t = l[off + k];
            int sh = fx_norm32(t);
            t = fx_shl(t, sh);
            t = (int32_t)(((int64_t)t * (int64_t)scale) >> 16);
            sh += scf_exp;
            if (sh < 0) t = fx_shl_sat(t, sh < -31 ? 31 : -sh);
            else t = fx_shr(t, sh > 31 ? 31 : sh);
            r[off + k] = t;
          }
        }
        off += width[sfb];
      }
      l += 128, r += 128;
    }
}

/* the reference's reciprocal square root and square root (basic_funcs.c:155-196) */
inline int32_t mul32_shl_sat(int32_t a, int32_t b) { /* basic_ops40.h: mult32_shl_sat */
  if (a == FX_MIN32 && b == FX_MIN32) return FX_MAX32;
  return fx_mul32_shl(a, b);
}
inline int32_t mul32x16_shl_sat(int32_t a, int16_t b) {
  if (a == FX_MIN32 && b == (int16_t)-32768) return FX_MAX32;
  return fx_mul32x16_shl(a, b);
}
inline int32_t mul32x16h_shl_sat(int32_t a, int32_t b) { /* basic_ops.h:62: the clamp looks at all of b */
  if (a == FX_MIN32 && b == -32768) return FX_MAX32;
  return fx_mul32x16_shl(a, (int16_t)(b >> 16));
}

int32_t one_by_sqrt(int32_t op) {
  int32_t a = fx_add_sat((int32_t)0x900ebee0, mul32x16_shl_sat(op, 0x39d9));
  int32_t iy = fx_add_sat(0x573b645a, mul32x16h_shl_sat(op, a));
  iy = fx_shl_dir_sat_limit(iy, 1);
  for (int it = 0; it < 3; it++) {
    a = mul32_shl_sat(op, iy);
    a = fx_sub_sat(0x40000000, fx_shl_dir_sat_limit(mul32_shl_sat(a, iy), 1));
    iy = fx_add_sat(iy, mul32_shl_sat(a, iy));
  }
  return iy;
}

int32_t fx_sqrt(int32_t op) {
  if (op == 0) return 0;
  int shift = fx_norm32(op) & ~1;
  op = fx_shl_dir_sat_limit(op, shift);
  shift = fx_shl_dir_sat_limit(shift, -1);
  op = mul32_shl_sat(one_by_sqrt(op), op);
  return fx_shl_dir_sat_limit(op, -(int)fx_sat16(shift - 1));
}

int32_t div32_pos_normb(int32_t a, int32_t b) { /* basic_ops.h:74-98: a / b in Q31 by 32 compare-subtract-shift steps */
  if (a == b) return FX_MAX32;
  uint32_t nr = (uint32_t)a, q = 0;
  const uint32_t dr = (uint32_t)b;
  for (int i = 0; i < 32; i++) {
    q <<= 1;
    if (nr >= dr) {
      nr -= dr;
      q += 1;
    }
    nr <<= 1;
  }
  return (int32_t)q;
}

void gen_rand_vec(int32_t scale, int shift, int32_t *x, int last, int32_t *seed) { /* pns_js_thumb.c:74-112 */
  int32_t nrg = 0;
  for (int i = 0; i <= last; i++) {
    *seed = (int32_t)((int64_t)1664525 * (int64_t)*seed + (int64_t)1013904223);
    x[i] = *seed >> 3;
    nrg = fx_add_sat(nrg, mul32_shl_sat(x[i], x[i]));
  }
  int nrg_scale = fx_norm32(nrg);
  if (nrg_scale > 0) {
    nrg_scale &= ~1;
    nrg = fx_shl_sat(nrg, nrg_scale);
    shift -= nrg_scale >> 1;
  }
  nrg = fx_sqrt(nrg);
  scale = div32_pos_normb(scale, nrg);
  if (shift < -31) shift = -31;
  for (int i = 0; i <= last; i++) x[i] = fx_shl_dir_sat_limit(mul32_shl_sat(x[i], scale), -shift);
}

void pns(XhCoreState *st, XhElement *el, int ch, int32_t *corr_seed) { /* pns_js_thumb.c:114-199 */
  XhChannel *c = &el->ch[ch];
  if (!c->pns_active) return;
  const XhIcs &ics = c->ics;
  const int16_t *swb = ics.window_sequence == XH_EIGHT_SHORT ? st->swb_short : st->swb_long;
  int32_t *spec = c->spec();
  for (int g = 0; g < ics.num_groups; g++)
    for (int w = 0; w < ics.group_len[g]; w++, spec += 128)
      for (int sfb = 0; sfb < ics.max_sfb; sfb++) {
        const int band = (g << 4) + sfb;
        if (!c->pns_used[band]) continue;
        const int last = swb[sfb + 1] - swb[sfb] - 1;
        const int32_t mant = xh_scale_mant_tab[c->sf[band] & 3];
        const int exp = 31 - (c->sf[band] >> 2) - 4; /* PNS_SCALE_MANT_TAB_SCALING -4 */
        int32_t *x = spec + swb[sfb];
        if (el->pns_correlated[band]) {
          if (ch == 0) {
            corr_seed[band] = st->pns_seed;
            gen_rand_vec(mant, exp, x, last, &st->pns_seed);
          } else {
            gen_rand_vec(mant, exp, x, last, &corr_seed[band]);
          }
        } else {
          gen_rand_vec(mant, exp, x, last, &st->pns_seed);
        }
      }
}

/* TNS, the 16-bit variant every stream of at most two channels takes (pns_js_thumb.c:248-514) */
void parcor_to_lpc(const int16_t *parcor, int16_t *lpc, int16_t *scale, int order) { /* aac_tns.c:147-202 */
  int status = 1;
  *scale = 0;
  while (status) {
    status = 0;
    int16_t t1[32 + 1] = {0}, t2[32 + 1] = {0};
    int32_t accu1 = 0x7fffffff >> *scale;
    for (int i = 0; i <= order; i++) {
      const int32_t accu = accu1;
      for (int j = 0; j < order; j++) {
        t2[j] = fx_round16(accu1);
        const int32_t prod = ((int32_t)parcor[j] * t1[j] == 0x40000000) ? FX_MAX32 : fx_shlw((int32_t)parcor[j] * t1[j], 1);
        accu1 = fx_add_sat(accu1, prod);
        if (fx_abs_sat(accu1) == 0x7fffffff) status = 1;
      }
      for (int j = order - 1; j >= 0; j--) {
        int32_t accu2 = fx_shlw((int32_t)t1[j], 16);
        const int32_t prod = ((int32_t)parcor[j] * t2[j] == 0x40000000) ? FX_MAX32 : fx_shlw((int32_t)parcor[j] * t2[j], 1);
        accu2 = fx_add_sat(accu2, prod);
        t1[j + 1] = fx_round16(accu2);
        if (fx_abs_sat(accu2) == 0x7fffffff) status = 1;
      }
      t1[0] = fx_round16(accu);
      lpc[i] = fx_round16(accu1);
      accu1 = 0;
    }
    if (status) *scale = (int16_t)(*scale + 1);
  }
}

/* One output of the all-pole filter: acc = sum over j = m .. 1 of mul32x16(s[i - j], lpc[j]), added up with saturation in
   that order (aac_tns.c:371-420).  If the magnitudes of the products add up to less than 2^31 no partial sum can leave
   the 32-bit range, the saturating chain is the plain sum and the order does not matter.  With L = sum |lpc[j]| and
   every state so far at most `quiet` = (2^31 - 1 - order) * 2^16 / L in magnitude that holds for sure
   (|floor(s * l / 2^16)| <= |s| |l| / 2^16 + 1): the case for every stream with the headroom the reference's scaling
   leaves (four bits), and it turns a chain of `order` dependent clamped adds per line into independent multiply-adds.
   From the first state beyond `quiet` on the chain is run as written. */
inline int32_t tns_acc_chain(const int32_t *h, const int16_t *lpc, int m) {
  int32_t acc = 0;
  for (int j = m; j > 0; j--) acc = fx_add_sat(acc, fx_mul32x16(h[-j], lpc[j]));
  return acc;
}
template <int M>
inline int32_t tns_acc_plain(const int32_t *h, const int16_t *lpc, int m_runtime) {
  const int m = M ? M : m_runtime;
  int64_t sum = 0;
  for (int j = 1; j <= m; j++) sum += ((int64_t)h[-j] * lpc[j]) >> 16; /* = fx_mul32x16, exactly */
  return (int32_t)sum;
}
inline uint32_t tns_mag(int32_t v) { return v < 0 ? 0u - (uint32_t)v : (uint32_t)v; }

/* lines from .. n-1; returns the first line it did not do (n, or where a state left the quiet range) */
template <int M>
inline int tns_ar_run(int32_t *x, int from, int n, int inc, const int16_t *lpc, int order, int shift_value, int scale_spec,
                      int32_t *hist, uint32_t quiet, uint32_t *loudest) {
  x += (ptrdiff_t)from * inc;
  uint32_t top = *loudest;
  int i = from;
  for (; i < n && top <= quiet; i++) {
    const int32_t y0 = fx_shl_sat(*x, scale_spec);
    const int32_t acc = tns_acc_plain<M>(hist + i, lpc, i < order ? i : order);
    /* y = sub_sat(y0, shl_sat(acc, 1)), state = shl_sat(y, shift_value): line i + 1 waits for this state, so the three
       clamps are first assumed idle (plain 64-bit arithmetic, checked beside the chain) and only redone if one was not */
    const int64_t t64 = 2 * (int64_t)acc, y64 = (int64_t)y0 - t64, s64 = (int64_t)((uint64_t)y64 << shift_value);
    int32_t y = (int32_t)y64, s = (int32_t)s64; /* the reference's state[0]: state[j] of step i is hist[i - 1 - j] */
    if (__builtin_expect(t64 != (int32_t)t64 || y64 != (int32_t)y64 || s64 != (int32_t)s64, 0)) {
      y = fx_sub_sat(y0, fx_shl_sat(acc, 1));
      s = fx_shl_sat(y, shift_value);
    }
    hist[i] = s;
    const uint32_t a = tns_mag(s);
    top = a > top ? a : top;
    *x = y >> scale_spec;
    x += inc;
  }
  *loudest = top;
  return i;
}

void tns_ar_filter(int32_t *x, int size, int inc, int16_t *lpc, int order, int shift_value, int scale_spec) {
  /* aac_tns.c:371-420: the order is rounded up to a multiple of four with zero coefficients, and the first `order`
     lines are filtered whether the region has that many or not */
  int32_t hist[1024 + 64];
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
  int64_t l1 = 0;
  for (int j = 1; j <= order; j++) l1 += lpc[j] < 0 ? -(int64_t)lpc[j] : lpc[j];
  const int64_t q = l1 ? (((int64_t)FX_MAX32 - order) << 16) / l1 : (int64_t)0xffffffff;
  const uint32_t quiet = q > (int64_t)0xffffffff ? 0xffffffffu : (uint32_t)q;
  uint32_t loudest = 0;
  const int lead = order < n ? order : n;
  int i = tns_ar_run<0>(x, 0, lead, inc, lpc, order, shift_value, scale_spec, hist, quiet, &loudest); /* fewer than `order` states yet */
  if (i == lead) {
    switch (order) {
      case 4: i = tns_ar_run<4>(x, lead, n, inc, lpc, order, shift_value, scale_spec, hist, quiet, &loudest); break;
      case 8: i = tns_ar_run<8>(x, lead, n, inc, lpc, order, shift_value, scale_spec, hist, quiet, &loudest); break;
      case 12: i = tns_ar_run<12>(x, lead, n, inc, lpc, order, shift_value, scale_spec, hist, quiet, &loudest); break;
      default: i = tns_ar_run<0>(x, lead, n, inc, lpc, order, shift_value, scale_spec, hist, quiet, &loudest); break;
    }
  }
  x += (ptrdiff_t)i * inc;
  for (; i < n; i++) { /* a state beyond the quiet range: the chain as the reference runs it */
    int32_t y = fx_shl_sat(*x, scale_spec);
    const int32_t acc = tns_acc_chain(hist + i, lpc, i < order ? i : order);
    y = fx_sub_sat(y, fx_shl_sat(acc, 1));
    hist[i] = fx_shl_sat(y, shift_value);
    *x = y >> scale_spec;
    x += inc;
  }
}

void tns(const XhCoreState *st, XhChannel *c) {
  const XhIcs &ics = c->ics;
  const bool is_short = ics.window_sequence == XH_EIGHT_SHORT;
  const int max_bands = xh_tns_max_bands[2 * st->sr_index + (is_short ? 1 : 0)];
  const int16_t *swb = is_short ? st->swb_short : st->swb_long;
  int32_t *spec = c->spec();
  for (int win = 0; win < (is_short ? 8 : 1); win++)
    for (int f = 0; f < c->tns.n_filt[win]; f++) {
      const XhTnsFilter &flt = c->tns.f[win][f];
      if (flt.order <= 0) continue;
      int16_t parcor[32 + 1], lpc[32 + 4 + 1];
      const int16_t *tab = flt.resolution ? xh_tns_coef4 : xh_tns_coef3;
      for (int i = 0; i < flt.order; i++) parcor[i] = tab[flt.coef[i] + (flt.resolution ? 8 : 4)];
      int lo = flt.start_band < max_bands ? flt.start_band : max_bands;
      if (lo > ics.max_sfb) lo = ics.max_sfb;
      int hi = flt.stop_band < max_bands ? flt.stop_band : max_bands;
      if (hi > ics.max_sfb) hi = ics.max_sfb;
      const int start = swb[lo], stop = swb[hi], size = stop - start;
      if (size <= 0) continue;
      int16_t scale_lpc;
      parcor_to_lpc(parcor, lpc, &scale_lpc, flt.order);
      int32_t *region = spec + (win << 7) + start;
      int32_t m = 0;
      for (int i = 0; i < size; i++) m |= fx_abs_nrm(region[i]);
      int scale_spec = fx_norm32(m);
      int position;
      if (flt.direction == -1) {
        position = stop - 1;
        if ((win << 7) + position < flt.order) continue;
      } else {
        position = start;
        if ((win << 7) + position + flt.order > 1024) continue;
      }
      scale_spec = scale_spec - 4 - scale_lpc;
      int32_t *at = spec + (win << 7) + position;
      if (scale_spec > 0) {
        if (scale_spec > 31) scale_spec = 31;
        tns_ar_filter(at, size, flt.direction, lpc, flt.order, scale_lpc, scale_spec);
      } else {
        /* not enough headroom: lines down, filter, lines up again.  The reference takes the lines it shifts down
           from window 0 whatever the window is (`win >> 7`, pns_js_thumb.c:455) and shifts the filtered window up */
        int32_t *down = spec + start;
        scale_spec = -scale_spec;
        if (scale_spec > 31) scale_spec = 31;
        for (int i = 0; i < size; i++) down[i] >>= scale_spec;
        tns_ar_filter(at, size, flt.direction, lpc, flt.order, scale_lpc, 0);
        for (int i = 0; i < size; i++) region[i] = fx_shlw(region[i], scale_spec);
      }
    }
}

int skip_pce(XhBits *br) { /* program_config_element: read over it (ISO/IEC 14496-3 4.4.1.1) */
  br->get(4);                 /* element_instance_tag */
  br->get(2);                 /* object_type */
  br->get(4);                 /* sampling_frequency_index */
  const int front = (int)br->get(4), side = (int)br->get(4), back = (int)br->get(4), lfe = (int)br->get(2);
  const int assoc = (int)br->get(3), cc = (int)br->get(4);
  if (br->get1()) br->get(4); /* mono_mixdown */
  if (br->get1()) br->get(4); /* stereo_mixdown */
  if (br->get1()) br->get(3); /* matrix_mixdown */
  br->skip(5 * (size_t)(front + side + back) + 4 * (size_t)lfe + 4 * (size_t)assoc + 5 * (size_t)cc);
  br->align();
  const int comment = (int)br->get(8);
  br->skip(8 * (size_t)comment);
  return br->overrun ? XH_ERR_BITS : 0;
}

}  // namespace

int32_t xh_inverse_quant(int32_t magnitude, int *err) { return pow43(magnitude, err); }

int xh_core_init(XhCoreState *st, int sr_index) {
  static const bool books_built = (build_books(), true); /* once, also when the first callers are parser threads */
  (void)books_built;
  if (sr_index < 0 || sr_index > 11) return XH_ERR_UNSUPPORTED;
  /* initfuncs.c:222-300: which width table a sampling frequency index takes */
  static const int8_t *const long_tab[12] = {xh_sfb_96_1024, xh_sfb_96_1024, xh_sfb_64_1024, xh_sfb_48_1024,
                                             xh_sfb_48_1024, xh_sfb_32_1024, xh_sfb_24_1024, xh_sfb_24_1024,
                                             xh_sfb_16_1024, xh_sfb_16_1024, xh_sfb_16_1024, xh_sfb_8_1024};
  static const int8_t *const short_tab[12] = {xh_sfb_96_128, xh_sfb_96_128, xh_sfb_96_128, xh_sfb_48_128,
                                              xh_sfb_48_128, xh_sfb_48_128, xh_sfb_24_128, xh_sfb_24_128,
                                              xh_sfb_16_128, xh_sfb_16_128, xh_sfb_16_128, xh_sfb_8_128};
  memset(st, 0, sizeof(*st));
  st->sr_index = sr_index;
  st->width_long = long_tab[sr_index];
  st->width_short = short_tab[sr_index];
  int n = 0, at = 0;
  for (st->swb_long[0] = 0; st->width_long[n] != -1; n++) st->swb_long[n + 1] = (int16_t)(at += st->width_long[n]);
  st->num_swb_long = n;
  n = 0, at = 0;
  for (st->swb_short[0] = 0; st->width_short[n] != -1; n++) st->swb_short[n + 1] = (int16_t)(at += st->width_short[n]);
  st->num_swb_short = n;
  return 0;
}

int xh_parse_raw_data_block(XhCoreState *st, XhBits *br, XhElement *el, int stage) {
  const size_t block_start = br->pos;
  int prev = XH_ID_END, have_channels = 0;
  el->n_ch = 0;
  el->sbr_bytes = 0;
  el->sbr_ext_type = 0;
  for (;;) {
    if (br->left() < 3) return XH_ERR_BITS;
    const int id = (int)br->get(3);
    if (id == XH_ID_END) break;
    switch (id) {
      case XH_ID_SCE:
      case XH_ID_LFE:
      case XH_ID_CPE: {
        if (have_channels) return XH_ERR_UNSUPPORTED; /* a second channel element: beyond two channels */
        have_channels = 1;
        el->id = id;
        el->n_ch = id == XH_ID_CPE ? 2 : 1;
        el->tag = (int)br->get(4);
        el->common_window = 0;
        memset(el->ms_used, 0, sizeof(el->ms_used));
        memset(el->pns_correlated, 0, sizeof(el->pns_correlated));
        for (int c = 0; c < el->n_ch; c++) {
          el->ch[c].pns_active = 0;
          el->ch[c].noise_energy = 0;
          memset(el->ch[c].pns_used, 0, sizeof(el->ch[c].pns_used));
          /* (a filter's fields are written by read_tns before tns() reads them: n_filt says how many are valid) */
          el->ch[c].tns.present = 0;
          memset(el->ch[c].tns.n_filt, 0, sizeof(el->ch[c].tns.n_filt));
        }
        if (id == XH_ID_CPE) {
          el->common_window = br->get1();
          if (el->common_window) {
            const int e = read_ics(st, br, &el->ch[0].ics);
            if (e) return e;
            el->ch[1].ics = el->ch[0].ics;
            const XhIcs &ics = el->ch[0].ics;
            const int mask = (int)br->get(2); /* channel.c:566-600 */
            for (int g = 0; g < ics.num_groups; g++)
              for (int sfb = 0; sfb < ics.max_sfb; sfb++) el->ms_used[g][sfb] = mask == 1 ? (uint8_t)br->get1() : (mask ? 1 : 0);
          }
        }
        for (int c = 0; c < el->n_ch; c++) {
          const int e = read_channel_stream(st, br, el, c);
          if (e) return e;
        }
        if (stage >= 2) { /* channel.c:602-692 */
          if (el->n_ch == 2) {
            if (el->common_window) {
              if (el->ch[0].pns_active || el->ch[1].pns_active) { /* channel.c:702-725 */
                const XhIcs &ics = el->ch[0].ics;
                for (int g = 0; g < ics.num_groups; g++)
                  for (int sfb = 0; sfb < ics.max_sfb; sfb++)
                    if (el->ms_used[g][sfb]) {
                      const int band = (g << 4) + sfb;
                      el->pns_correlated[band] = 1;
                      if (el->ch[0].pns_used[band] && el->ch[1].pns_used[band]) el->ms_used[g][sfb] ^= 1;
                    }
              }
              ms_stereo(st, el);
            }
            intensity_stereo(st, el);
          }
          for (int c = 0; c < el->n_ch; c++) {
            pns(st, el, c, st->pns_corr_seed);
            if (el->ch[c].tns.present) tns(st, &el->ch[c]);
          }
        }
        break;
      }
      case XH_ID_DSE: { /* common_lpfuncs.c:175-235 */
        const uint32_t v = br->get(13);
        int cnt = (int)(v & 0xff);
        if (cnt == 255) cnt += (int)br->get(8);
        if (v & 0x100) br->skip((8 - ((br->pos - block_start) & 7)) & 7);
        br->skip(8 * (size_t)cnt);
        break;
      }
      case XH_ID_PCE: {
        const int e = skip_pce(br);
        if (e) return e;
        break;
      }
      case XH_ID_FIL: { /* aacpluscheck.c:59-265 */
        int count = (int)br->get(4);
        if (count == 15) count = (int)br->get(8) + 14;
        if (count > 0) {
          const int type = (int)br->get(4);
          if ((type == 13 || type == 14) && (prev == XH_ID_SCE || prev == XH_ID_CPE) && count <= (int)sizeof(el->sbr) &&
              el->sbr_bytes == 0) {
            el->sbr_ext_type = type;
            el->sbr_bytes = count;
            el->sbr[0] = (uint8_t)br->get(4);
            int i = 1;
            for (; i + 4 <= count; i += 4) { /* four payload bytes per look (they sit at any bit offset) */
              const uint32_t w = br->peek32();
              el->sbr[i] = (uint8_t)(w >> 24), el->sbr[i + 1] = (uint8_t)(w >> 16), el->sbr[i + 2] = (uint8_t)(w >> 8), el->sbr[i + 3] = (uint8_t)w;
              br->skip(32);
            }
            for (; i < count; i++) el->sbr[i] = (uint8_t)br->get(8);
          } else {
            br->get(4);
            br->skip(8 * (size_t)(count - 1));
          }
        }
        break;
      }
      default:
        return XH_ERR_UNSUPPORTED; /* CCE */
    }
    if (br->overrun) return XH_ERR_BITS;
    prev = id;
  }
  br->skip((8 - ((br->pos - block_start) & 7)) & 7);
  if (br->overrun) return XH_ERR_BITS;
  return have_channels ? 0 : XH_ERR_SYNTAX;
}
