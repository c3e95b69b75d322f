/*
 * aac_core.h -- host-side AAC-LC syntax decoder: one raw_data_block (ISO/IEC 14496-3 4.4.2.1 as the reference reads it,
 * decoder/ixheaacd_aacdecoder.c:362-647) -> the spectra of its SCE / CPE exactly as the reference hands them to
 * ixheaacd_imdct_process (Q-format, rounding and the reference's quirks included), plus the raw SBR extension payload.
 * CPU code: the bitstream syntax is serial; everything behind this seam runs on the GPU.
 * Scope: AAC-LC objects (AOT 2; 5 / 29 through the SBR payload), 1024-line frames, one SCE or CPE per raw_data_block
 * (at most two channels: the reference's q_factor / TNS variants for more than two channels are not built), no LTP, no
 * gain control, no DRC payload handling (read and ignored), no error concealment.
 */
#ifndef XAAC_HOST_AAC_CORE_H
#define XAAC_HOST_AAC_CORE_H

#include <stdint.h>

#include "bits.h"

enum { XH_ONLY_LONG = 0, XH_LONG_START = 1, XH_EIGHT_SHORT = 2, XH_LONG_STOP = 3 };
enum { XH_ID_SCE = 0, XH_ID_CPE = 1, XH_ID_CCE = 2, XH_ID_LFE = 3, XH_ID_DSE = 4, XH_ID_PCE = 5, XH_ID_FIL = 6, XH_ID_END = 7 };
enum { XH_ZERO_HCB = 0, XH_ESC_HCB = 11, XH_NOISE_HCB = 13, XH_INTENSITY_HCB2 = 14, XH_INTENSITY_HCB = 15 };

/* error codes (negative returns) */
enum {
  XH_ERR_BITS = -1,        /* ran out of bits */
  XH_ERR_SYNTAX = -2,      /* a value the syntax forbids (max_sfb, section length, code book 12, pulse / TNS range ...) */
  XH_ERR_UNSUPPORTED = -3, /* prediction, gain control, CCE, more elements than this decoder builds */
  XH_ERR_ESCAPE = -4       /* escape value beyond what the inverse quantiser takes (channel.c:1066) */
};

struct XhIcs {
  int window_sequence, window_shape, max_sfb, num_swb, num_groups;
  uint8_t group_len[8];
};

struct XhTnsFilter {
  int start_band, stop_band, order, direction, resolution;
  int8_t coef[32];
};

struct XhTns {
  int present;
  int n_filt[8];
  XhTnsFilter f[8][4];
};

struct XhPulse {
  int present, number, start_band;
  uint8_t offset[4], amp[4];
};

#define XH_SPEC_SLACK 16 /* the reference's TNS filter runs order (rounded up to 4) lines even over a shorter region */

struct XhChannel {
  XhIcs ics;
  int global_gain;
  uint8_t cb[8 * 16];  /* long blocks: bands 0 .. 50 contiguous; short: 16 per group (longblock.c:146) */
  int16_t sf[8 * 16];
  XhPulse pulse;
  XhTns tns;
  int pns_active;
  int16_t noise_energy;
  uint8_t pns_used[8 * 16];
  /* the frame's lines: XH_SPEC_WORDS words of the CALLER's (set before xh_parse_raw_data_block; a parser that keeps one
     such buffer per thread instead of one per stream has 8 KB less state to drag through the caches per frame) */
  int32_t *spec_mem;
  int32_t *spec() { return spec_mem + XH_SPEC_SLACK; }
};
#define XH_SPEC_WORDS (1024 + 2 * XH_SPEC_SLACK)

/* what is constant between ADTS headers of one stream, and the little state that outlives a frame */
struct XhCoreState {
  int sr_index;            /* 0 .. 11 */
  int16_t swb_long[52], swb_short[16];
  const int8_t *width_long, *width_short;
  int num_swb_long, num_swb_short;
  int32_t pns_seed;        /* pstr_pns_rand_vec_data->current_seed: starts at 0, runs on from frame to frame */
  int32_t pns_corr_seed[8 * 16]; /* pstr_pns_corr_info->random_vector: the left channel's seed of a band whose noise the M/S
                                    flag correlates; a right channel that substitutes noise in such a band where the left one
                                    does not finds the value an earlier frame left (the reference keeps it in scratch memory
                                    that AAC-LC decoding does not reuse in between) */
};

struct XhElement {
  int id;                   /* XH_ID_SCE / XH_ID_CPE */
  int n_ch, tag, common_window;
  uint8_t ms_used[8][64];
  uint8_t pns_correlated[8 * 16];
  XhChannel ch[2];
  /* SBR extension payload of the FIL element behind the channel element (aacpluscheck.c:59): extension type 13 / 14,
     then the payload bytes with the first byte holding the 4 bits that follow the extension type */
  int sbr_ext_type, sbr_bytes;
  uint8_t sbr[272];
};

/* |q|^(4/3) in Q13 of a quantised magnitude as the reference computes it (table, then its interpolation: channel.c:1055);
   *err is set beyond 8191 + 32 */
int32_t xh_inverse_quant(int32_t magnitude, int *err);

/* sr_index 0 .. 11; returns 0 or XH_ERR_UNSUPPORTED */
int xh_core_init(XhCoreState *st, int sr_index);

/* Parses one raw_data_block at the reader's position (up to and including ID_END and the byte alignment) into the line
   buffers el->ch[c].spec_mem point to, dequantises,
   applies the scale factors and the tools (M/S, intensity, PNS, TNS): el->ch[c].spec() are the lines the IMDCT takes.
   `stage`: 2 = everything; 1 = stop before the tools (spectra as at the entry of ixheaacd_channel_pair_process).
   Returns 0 or a negative XH_ERR_*. */
int xh_parse_raw_data_block(XhCoreState *st, XhBits *br, XhElement *el, int stage);

#endif /* XAAC_HOST_AAC_CORE_H */
