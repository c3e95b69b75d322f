/*
 * xaac_parse.h -- C ABI of the host-side bitstream front end (libxaac_amd/libxaac_host.so): ADTS framing and the
 * AAC-LC / SBR / PS syntax, decoded on the CPU into exactly the frame-level inputs the GPU entry points of xaac_amd.h
 * take (xaac_imdct_batch spectra and window info; xaac_sbr_header / xaac_sbr_frame / xaac_ps_frame side info).
 * It replaces, for real streams, what a reference decoder instance does between its input buffer and those seams:
 *   ixheaacd_adtsframe                  decoder/ixheaacd_headerdecode.c:316
 *   ixheaacd_aacdec_decodeframe         decoder/ixheaacd_aacdecoder.c:100   (element loop :362-647)
 *   ixheaacd_individual_ch_stream       decoder/ixheaacd_channel.c:481, ixheaacd_channel_pair_process :602
 *   ixheaacd_check_for_sbr_payload      decoder/ixheaacd_aacpluscheck.c:59
 *   ixheaacd_sbr_read_sce / _cpe        decoder/ixheaacd_env_extr.c, ixheaacd_dec_sbrdata decoder/ixheaacd_env_dec.c
 *   ixheaacd_read_ps_data / ixheaacd_decode_ps_data   decoder/ixheaacd_ps_bitdec.c
 * No reference code runs in the process.  CPU only: no GPU or torch dependency.
 */
#ifndef XAAC_PARSE_H
#define XAAC_PARSE_H

#include <stddef.h>
#include <stdint.h>

#include "xaac_esbr.h"
#include "xaac_sbr.h"

#ifdef __cplusplus
extern "C" {
#endif

#define XAAC_PARSE_OK 0
#define XAAC_PARSE_NEED_DATA 1        /* fewer bytes than one whole ADTS frame */
#define XAAC_PARSE_ERR_SYNC -10       /* no ADTS sync word at the given position */
#define XAAC_PARSE_ERR_HEADER -11     /* profile other than AAC-LC, layer != 0, sampling index > 11, frame length < 8 */
#define XAAC_PARSE_ERR_BITS -1        /* the frame ran out of bits */
#define XAAC_PARSE_ERR_SYNTAX -2      /* a value the syntax forbids */
#define XAAC_PARSE_ERR_UNSUPPORTED -3 /* outside this front end's scope (see libxaac_amd/host/aac_core.h) */
#define XAAC_PARSE_ERR_ESCAPE -4      /* spectral escape value beyond the inverse quantiser's range */

#define XAAC_TOOL_MS 1
#define XAAC_TOOL_INTENSITY 2
#define XAAC_TOOL_PNS 4
#define XAAC_TOOL_TNS 8
#define XAAC_TOOL_PULSE 16
#define XAAC_TOOL_SHORT 32
#define XAAC_TOOL_ESCAPE 64

typedef struct xaac_adts_header {
  int32_t id, layer, protection_absent, profile; /* profile = 2 bit field + 1, as the reference counts it */
  int32_t sr_index, sampling_rate, channel_config, frame_bytes, raw_blocks, header_bytes;
} xaac_adts_header;

/* one decoded AAC-LC core frame: what ixheaacd_imdct_process (decoder/ixheaacd_lpfuncs.c:347) is handed per channel */
typedef struct xaac_core_frame {
  int32_t n_ch;            /* 1 (SCE) or 2 (CPE) */
  int32_t element_id;      /* 0 SCE, 1 CPE, 3 LFE */
  int32_t common_window;
  int32_t sbr_ext_type;    /* 0: no SBR payload; 13 SBR_EXTENSION, 14 SBR_EXTENSION_CRC */
  int32_t sbr_bytes;       /* bytes of sbr[] (first byte = the 4 bits behind the extension type) */
  int32_t tools;           /* which tools the frame used: XAAC_TOOL_* bits (informative; tests assert coverage with it) */
  struct {
    int16_t window_sequence, window_shape, max_sfb, num_window_groups;
  } ics[2];
  int32_t spec[2][1024];   /* spectral lines in the reference's Q-format; short frames: 8 x 128, window by window */
  uint8_t sbr[272];
} xaac_core_frame;

/* the SBR (+ PS) side info of the frame, decoded as the reference's -esbr:0 path decodes it (ixheaacd_applysbr,
   decoder/ixheaacd_sbrdecoder.c:313-760), and what the frame asks of the host beside handing the side info on */
typedef struct xaac_sbr_side {
  int32_t apply;          /* sync_state == SBR_ACTIVE: frame[].apply_processing = 1 */
  int32_t reset;          /* ixheaacd_sbr_dec_reset ran (sbrdecoder.c:103-252): before this frame's GPU call set, in the
                             xaac_sbr_state of channels 0 .. reset_channels-1: ph_index = 0, filt_buf_noise_e = 0,
                             start_up = 1, bw_array_prev = 0, syn_lsb = codec_usb = header.sub_band_start,
                             syn_usb = header.sub_band_end */
  int32_t reset_channels;
  int32_t upsampling;     /* ixheaacd_prepare_upsamp ran (:254): syn_lsb = codec_usb = 32, syn_usb = 64 in those channels */
  int32_t stereo;         /* channel pair: frame[0] and frame[1] are both in use */
  int32_t ps;             /* header.channel_mode == PS_STEREO: ps_frame is valid, the output has two channels */
  int32_t ps_start;       /* first PS frame behind mono frames: xaac_sbr_state_handover(XAAC_HANDOVER_PS_START) first (:762) */
  int32_t frame_ok;       /* the payload parsed and its length / CRC checked out */
  xaac_sbr_header header; /* one per stream: the channels of a pair share it */
  xaac_sbr_frame frame[2];
  xaac_ps_frame ps_frame;
} xaac_sbr_side;

typedef struct xaac_parser xaac_parser;

XAAC_API int32_t xaac_parser_create(xaac_parser **p);
XAAC_API void xaac_parser_destroy(xaac_parser *p);

/* The ADTS header at data[0 .. n): XAAC_PARSE_OK, _NEED_DATA (n < 7 / 9), _ERR_SYNC or _ERR_HEADER. */
XAAC_API int32_t xaac_adts_parse_header(const uint8_t *data, size_t n, xaac_adts_header *h);

/* Decodes the next raw_data_block -- one frame of 1024 samples per channel -- of the ADTS stream at data[0 .. n) into `out`.
   stage 2: spectra as the IMDCT takes them; stage 1: as they are before the M/S, intensity, PNS and TNS tools (the entry
   of ixheaacd_channel_pair_process).  *consumed = what the call used up: the ADTS frame's length, or -- in a frame with
   number_of_raw_data_blocks_in_frame > 0 (headerdecode.c:353, api.c:2909-2925) -- header + first block for the first call and
   one block (+ its CRC word in a protected frame) for each of the following ones, which must be handed the bytes right
   behind what the call before consumed.  The parser keeps what outlives a call (the blocks left in the frame, PNS random
   seed, SBR / PS decoding state).  Protected frames (protection_absent == 0) with several blocks follow ISO/IEC 13818-7 --
   a crc_check word behind every block -- which deliberately differs from the reference: its api.c:3760-3767 never skips
   those words in the follow-up calls (its per-call `adts` struct is zero there) and misparses blocks 2..N. */
XAAC_API int32_t xaac_parse_adts_frame(xaac_parser *p, const uint8_t *data, size_t n, int32_t stage, xaac_core_frame *out,
                              size_t *consumed);

/* The SBR / PS side info of the frame xaac_parse_adts_frame decoded last (its payload is in the parser; a frame without
   one counts as a frame whose SBR data is missing, as in the reference).  ps_enable: parametric stereo allowed (mono
   streams).  The first call fixes the stream's SBR configuration (output rate = twice the core rate).  Returns
   XAAC_PARSE_OK, or XAAC_PARSE_ERR_SYNTAX where the reference returns a fatal error from ixheaacd_applysbr. */
XAAC_API int32_t xaac_parse_sbr_side(xaac_parser *p, int32_t ps_enable, xaac_sbr_side *side);

/* The reference's default interpretation of the SBR payload (-esbr:1, "Path A": decoder/ixheaacd_sbrdecoder.c:479-493 the
   payload runs one frame late; the ENHSBR extension element with patching mode / pitch, env_extr.c:595-714; scale factors
   and noise floors handed to the float tools, env_dec.c:52-72, :586-626) instead of the -esbr:0 one.  To be chosen before
   the stream's first xaac_parse_sbr_side call; esbr 0 / 1. */
XAAC_API int32_t xaac_parser_set_esbr(xaac_parser *p, int32_t esbr);
/* ... and for such a stream, after xaac_parse_sbr_side: the xaac_esbr_side of channel 0 / 1 of the frame (what
   xaac_esbr_sbr_process_batch takes beside header and frame) */
XAAC_API int32_t xaac_parse_esbr_side(xaac_parser *p, int32_t channel, xaac_esbr_side *side);
/* ... and for a frame with side.reset: the pitch_in_bins that ixheaacd_sbr_dec_reset hands to its two transposer runs (the
   first channel's, from the payload before this frame's: sbrdecoder.c:547-550) */
XAAC_API int32_t xaac_parse_reset_pitch(xaac_parser *p, int32_t *pitch_in_bins);

/* The inverse quantiser of spectral magnitudes, |q|^(4/3) in Q13, exactly as the reference computes it (table up to 128, its
   linear interpolation beyond, decoder/ixheaacd_channel.c:1055-1093; _ERR_ESCAPE past 8191 + 32).  Exposed for tests. */
XAAC_API int32_t xaac_inverse_quant(int32_t magnitude, int32_t *out);

/* ---- one frame of N streams at once ----------------------------------------------------------------------------------
 * What a batched host runs per step: every stream's next ADTS frame parsed (core, and SBR / PS side info when with_sbr)
 * on a team of CPU threads, the results written straight into the host staging arrays -- pinned, in the layouts the GPU
 * entry points take -- that one cudaMemcpyAsync per array then moves.  Streams are independent, so is their parsing. */
typedef struct xaac_parse_batch {
  int32_t n_streams;
  int32_t n_ch;               /* core channels of every stream (1 or 2) */
  int32_t with_sbr;           /* also decode the SBR / PS side info (ps_enable as in xaac_parse_sbr_side) */
  int32_t ps_enable;
  int32_t stage;              /* as in xaac_parse_adts_frame */
  int32_t threads;            /* worker threads (the caller counts as one), <= 0: half the hardware threads, at most 48 and at most
                                 twice what the scheduler affinity and the cgroup CPU quota grant the process */
  xaac_parser *const *parser; /* [n_streams] */
  const uint8_t *const *data; /* [n_streams] the frame's first byte */
  const uint64_t *bytes;      /* [n_streams] bytes available there */
  int32_t *spec;              /* [n_streams][n_ch][1024] */
  uint8_t *ics;               /* [n_streams][n_ch][2]: window_sequence, window_shape (xaac_ics_info) */
  xaac_sbr_header *header;    /* with_sbr: [n_streams][n_ch] (the channels of a pair get copies) */
  xaac_sbr_frame *frame;      /* with_sbr: [n_streams][n_ch] */
  xaac_ps_frame *ps_frame;    /* with_sbr, optional: [n_streams] */
  int32_t *flags;             /* with_sbr: [n_streams][8] = apply, reset, reset_channels, upsampling, stereo, ps, ps_start,
                                 frame_ok of xaac_sbr_side */
  int32_t *tools;             /* optional [n_streams] */
  uint64_t *consumed;         /* [n_streams] frame length (0 where status != 0) */
  int32_t *status;            /* [n_streams] XAAC_PARSE_OK / _NEED_DATA / error: such a stream's rows are left as they were */
  xaac_esbr_side *esbr_side;  /* with_sbr, parsers in xaac_parser_set_esbr(1) mode, optional: [n_streams][n_ch] */
  int32_t *reset_pitch;       /* ... optional: [n_streams], written for frames with flags[1] (reset): xaac_parse_reset_pitch */
  uint64_t *pos;              /* optional [n_streams], in / out: the library keeps the streams' read positions -- stream i's frame
                                 starts at data[i] + pos[i] with bytes[i] - pos[i] bytes available, and pos[i] moves on by the
                                 frame's length where status is XAAC_PARSE_OK.  A host that leaves data / bytes as the whole
                                 streams can then issue the next step's call without touching its arrays in between.
                                 (Appended in round 4, like the two members below: see xaac_parse_batch_run_sized.) */
  int32_t frames;             /* 0 or 1: one frame per stream and call.  T > 1 (needs pos): up to T consecutive frames of every
                                 stream per call -- a stream's parser state and bytes are fetched once for T frames -- with every
                                 output array T times as long, step t's rows behind step t - 1's (spec [T][n_streams][n_ch][1024],
                                 status [T][n_streams], flags [T][n_streams][8], ...; consumed [n_streams] = the bytes of all its
                                 frames).  A stream that runs out or fails at step t has that status word in steps t .. T - 1.
                                 The call then returns the number of frames parsed. */
  int32_t *lines;             /* optional [frames][n_streams], out: for a delivered frame the number of leading spectral lines
                                 (a multiple of 16, the larger of the stream's channels) behind which every line is zero -- what
                                 a host needs to send up of the frame's rows (AAC + SBR streams code the lower half of the
                                 spectrum or less) */
} xaac_parse_batch;

/* returns the number of streams whose status is XAAC_PARSE_OK, or a negative XAAC_PARSE_ERR_* for a bad descriptor */
XAAC_API int32_t xaac_parse_batch_run(const xaac_parse_batch *b);
/* The same call in two halves, for a host whose calling thread has other work meanwhile (queueing the copies and launches
   of the step before): _start hands the batch to the worker team (all `threads` of them team threads: the caller does not
   parse along) and returns at once; _wait blocks until every stream is parsed and returns what _run would have.  The
   descriptor is copied; the arrays it points to belong to the team until _wait returns.  One batch in flight per process:
   every _start needs its _wait (on any thread) before the next _start or _run gets the team; _wait without a _start returns
   XAAC_PARSE_ERR_SYNTAX.  busy_seconds (optional): how long the team parsed, from _start until its last thread ran out of
   streams -- what the caller overlapped, or waited for. */
XAAC_API int32_t xaac_parse_batch_start(const xaac_parse_batch *b);
XAAC_API int32_t xaac_parse_batch_wait(double *busy_seconds);
/* The descriptor has grown at its end (round 4: pos, frames, lines) and may again.  _run_sized / _start_sized take the size of
   the caller's struct: members behind it read as zero, whatever lies behind the caller's shorter struct is not looked at.
   The symbols xaac_parse_batch_run / _start themselves keep the ORIGINAL layout's meaning -- they read the descriptor up to and
   including reset_pitch, as a binary built before the growth expects --; code compiled against this header reaches the sized
   entry points with its own sizeof through the two macros below (define XAAC_PARSE_NO_SIZED_MACROS to call the symbols). */
XAAC_API int32_t xaac_parse_batch_run_sized(const xaac_parse_batch *b, uint64_t struct_size);
XAAC_API int32_t xaac_parse_batch_start_sized(const xaac_parse_batch *b, uint64_t struct_size);
#ifndef XAAC_PARSE_NO_SIZED_MACROS
#define xaac_parse_batch_run(b) xaac_parse_batch_run_sized((b), sizeof(xaac_parse_batch))
#define xaac_parse_batch_start(b) xaac_parse_batch_start_sized((b), sizeof(xaac_parse_batch))
#endif

/* ---- the states of a new stream, and the frame-level state changes of ixheaacd_applysbr -------------------------------
 * Host-side helpers on HOST copies of the boundary structs (the host writes them to the device once per stream, and on
 * the rare frames with side->reset / side->upsampling reads the stream's state back, applies the change, writes it again). */
/* ixheaacd_init_sbr for one channel: decoder/ixheaacd_sbrdec_initfuncs.c:1133-1135 (bank scales), :1235-1238 (scale
   factors), :870-898 (envelope calculator, previous-frame data) */
XAAC_API void xaac_sbr_state_init(xaac_sbr_state *s);
/* ... and the parametric stereo tool with the right channel's bank: :1050-1059 */
XAAC_API void xaac_ps_state_init(xaac_ps_state *s);
/* the Path A (-esbr:1) states of a new stream: zeros but esbr_start_up = 1 (sbrdec_initfuncs.c:1018) and the PS mixing
   matrix's h11 / h12 real parts = 1.0 (ps_dec_flt.c:349-352); the transposer's parameters are zero until the first reset */
XAAC_API void xaac_esbr_state_init(xaac_esbr_state *s);
XAAC_API void xaac_esbr_ps_state_init(xaac_esbr_ps_state *s);
XAAC_API void xaac_hbe_state_init(xaac_hbe_state *s);
/* ixheaacd_qmf_hbe_data_reinit (decoder/ixheaacd_hbe_trans.c:102-222) as ixheaacd_sbr_dec_reset calls it for a 2:1 stream
   with 1024-line core frames: the bank size, first band, band range and cross-over bands of the QMF transposer from the
   header's band tables; clears the two banks' delay lines.  `s` is the channel's state as it is (a new stream's, or the one a
   header before this one left: max_stretch and fft_ready carry over where the reference leaves them alone).  Returns 0, or
   -1 where the reference returns an error.
   (The reset's two transposer runs over the rows the channel holds, sbrdecoder.c:196-236, are the caller's:
   xaac_hbe_apply_batch on the device-resident state.) */
XAAC_API int32_t xaac_hbe_state_reinit(xaac_hbe_state *s, const xaac_sbr_header *header);
/* ... for n channels at once, on the states' integer tails alone (everything from synth_size on: the members the re-initialisation
   computes; a batched host keeps them beside the device-resident states and clears the two delay lines on the device):
   tails [n][XAAC_HBE_TAIL_BYTES] in / out, headers [n].  Returns -1, or the index of the first channel whose band tables the
   reference would refuse (the tails behind it are left as they were). */
#define XAAC_HBE_TAIL_BYTES (sizeof(xaac_hbe_state) - offsetof(xaac_hbe_state, synth_size))
XAAC_API int32_t xaac_hbe_state_reinit_tails(uint8_t *tails, const xaac_sbr_header *headers, int32_t n);
/* ixheaacd_dft_hbe_data_reinit (decoder/ixheaacd_hbe_dft_trans.c:272-455; -esbr_hq:1) for a 2:1 stream with 1024-line core frames:
   the DFT transposer's sizes, band range and cross-over bands into `s` (its synthesis bank's delay line cleared, :302;
   max_stretch carries over where the reference leaves it alone), the two time windows and the patches' cross-over windows into
   `cfg`, the analysis bank's coefficient matrices into coef_re / coef_im ([64][128] floats each: :374-388, libm cos / sin as
   the reference calls them).  Returns 0, or -1 where the reference returns an error or the sizes have no room in the structs
   (xaac_hbe.h: sizes the reference has transforms for always fit). */
XAAC_API int32_t xaac_hbe_dft_state_reinit(xaac_hbe_dft_state *s, xaac_hbe_dft_cfg *cfg, float *coef_re, float *coef_im,
                                           const xaac_sbr_header *header);
/* what ixheaacd_sbr_dec_reset (sbrdecoder.c:103-252) and ixheaacd_prepare_upsamp (:254-276) do to one channel's state
   for this frame's side info; channel = 0 or 1 (no-op beyond side->reset_channels / for frames without either) */
XAAC_API void xaac_sbr_state_apply_side(xaac_sbr_state *s, const xaac_sbr_side *side, int32_t channel);
/* the same for the right channel's synthesis bank kept in the PS state (channel 1 of a mono + PS stream) */
XAAC_API void xaac_ps_state_apply_side(xaac_ps_state *s, const xaac_sbr_side *side);

#ifdef __cplusplus
}
#endif

#endif /* XAAC_PARSE_H */
