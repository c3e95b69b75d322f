/*
 * xaac_amd.h -- C ABI of libxaac_amd.so: the MI355X (gfx950) back-end for the
 * transform hot path of the libxaac decoder.
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes plain
 * pointers and sizes, allocates nothing behind the caller's back (all buffers
 * are caller-owned, like the reference's memtab protocol, README_dec.md:44-55)
 * and returns an IA_ERRORCODE-style int32: 0 = OK, bit 31 set = fatal
 * (common/ixheaac_error_standards.h:23-25).
 *
 * Reference seam replaced (see INTEGRATION.md for the reference-side stub):
 *   xaac_imdct_process_batch  <->  ixheaacd_imdct_process
 *        decl decoder/ixheaacd_block.h:132, def decoder/ixheaacd_lpfuncs.c:347-802,
 *        sole core call site decoder/ixheaacd_aacdecoder.c:988, plus the
 *        WORD32->WORD16 hand-off that follows it
 *        (decoder/ixheaacd_api.c:337-370 SBR case; decoder/ixheaacd_peak_limiter.c:324
 *        + decoder/ixheaacd_api.c:3676-3681 AAC-LC case with -peak_limiter_off:1).
 *   The per-op function pointers it subsumes are the ones the reference's
 *   selector binds at decoder/x86/ixheaacd_function_selector_x86.c:89-125
 *   (calc_max_spectral_line, pretwiddle_compute, imdct_using_fft, post_twiddle,
 *   post_twid_overlap_add, over_lap_add1/2, spec_to_overlapbuf, overlap_buf_out,
 *   overlap_out_copy, neg_shift_spec).
 *
 * One call processes N independent channel-frames ("channel-frame" = one
 * channel of one 1024-sample access unit).  The reference processes one per
 * call; the batch is the only change of shape.
 */
#ifndef XAAC_AMD_H
#define XAAC_AMD_H

#include <stdint.h>

#include "xaac_sbr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (bit 31 = fatal, as in the reference) ------------------- */
#define XAAC_OK 0
#define XAAC_FATAL_NULL_ARG ((int32_t)0xFFFF8000)
#define XAAC_FATAL_BAD_ARG ((int32_t)0xFFFF8001)
#define XAAC_FATAL_NO_DEVICE ((int32_t)0xFFFF8002)
#define XAAC_FATAL_HIP ((int32_t)0xFFFF8003)
#define XAAC_FATAL_BAD_WINDOW_SEQ ((int32_t)0xFFFF8004)

/* window_sequence / window_shape values: decoder/ixheaacd_cnst.h:100-103 */
enum { XAAC_ONLY_LONG = 0, XAAC_LONG_START = 1, XAAC_EIGHT_SHORT = 2, XAAC_LONG_STOP = 3 };
enum { XAAC_WIN_SINE = 0, XAAC_WIN_KBD = 1 };

/* PCM16 hand-off flavour fused behind the IMDCT */
enum {
  XAAC_PCM_LC = 0, /* x * 2^qshift_adj wrapping, then round16: ixheaacd_scale_adjust + api.c:3676-3681 */
  XAAC_PCM_SBR = 1 /* round16(shl32_sat(x, qshift_adj)): api.c:353-366 (core -> SBR hand-off) */
};

/* The two fields of ia_ics_info_struct (decoder/ixheaacd_channelinfo.h:33-47)
 * the path reads; frame_length is fixed at 1024. */
typedef struct xaac_ics_info {
  uint8_t window_sequence;
  uint8_t window_shape;
} xaac_ics_info;

/* Per-channel persistent IMDCT state besides the overlap samples: the
 * window_shape / window_sequence members of ia_aac_dec_overlap_info
 * (decoder/ixheaacd_channelinfo.h:88-100), updated by every call
 * (lpfuncs.c:800-801). Zero-initialised for a new stream. */
typedef struct xaac_ovl_state {
  uint8_t window_sequence;
  uint8_t window_shape;
} xaac_ovl_state;

/* Batch descriptor.  All pointers are DEVICE pointers for
 * xaac_imdct_process_batch and HOST pointers for ..._batch_host.
 * Channel-frame i belongs to access unit i / ch_fac, channel i % ch_fac, and
 * its output sample n lands at index (i / ch_fac) * 1024 * ch_fac + n * ch_fac
 * + i % ch_fac (the reference's interleaved `out_samples` at stride ch_fac). */
typedef struct xaac_imdct_batch {
  int32_t n_ch;               /* number of channel-frames, >= 0; multiple of ch_fac */
  int32_t ch_fac;             /* output interleave stride, 1 or 2 */
  const int32_t *spec;        /* [n_ch][1024]  spectral coefficients (not modified) */
  const xaac_ics_info *ics;   /* [n_ch] */
  int32_t *overlap;           /* [n_ch][512]   in/out: ptr_overlap_buf */
  xaac_ovl_state *state;      /* [n_ch]        in/out */
  int32_t *out32;             /* optional [n_ch*1024] WORD32 time samples (reference boundary) */
  int16_t *pcm16;             /* optional [n_ch*1024] PCM16 after the qshift_adj hand-off */
  int8_t *qshift_adj;         /* optional [n_ch] ics->qshift_adj as the reference sets it */
  int32_t pcm_mode;           /* XAAC_PCM_LC or XAAC_PCM_SBR (used when pcm16 != NULL) */
  int32_t *status;            /* optional [n_ch]: XAAC_OK, or XAAC_FATAL_BAD_WINDOW_SEQ for a channel-frame whose
                                 window_sequence > 3 / window_shape > 1 (in ics or in state) -- values the 2-bit / 1-bit
                                 bitstream fields cannot carry; such a channel-frame is left untouched */
} xaac_imdct_batch;

/* ---- AAC-LD / AAC-ELD IMDCT ------------------------------------------------------------
 * xaac_imdct_ld_process_batch <-> ixheaacd_imdct_process with ics->frame_length 512 or 480 and object_type AOT_ER_AAC_LD (23)
 * or AOT_ER_AAC_ELD (39) (decoder/ixheaacd_lpfuncs.c:385-409, :456-486): ixheaacd_inverse_transform_512 / ixheaacd_mdct_480_ld
 * (aac_imdct.c:1761 / :1707), then ixheaacd_lap1_512_480 (block.c:1140) for LD or the sign / copy step and
 * ixheaacd_eld_dec_windowing (lpfuncs.c:804) for ELD.  ONLY_LONG frames (the two profiles have no others); PCM16 out, which
 * is what the reference writes here (qshift_adj = -2); ld_mps_present = 0, slot_element = 0. */
typedef struct xaac_imdct_ld_batch {
  int32_t n_ch;                 /* channel-frames; a multiple of ch_fac */
  int32_t ch_fac;               /* output interleave stride, 1 or 2 */
  int32_t frame_length;         /* 512 or 480 */
  int32_t eld;                  /* 0: AAC-LD (sine / low-overlap window by the PREVIOUS frame's shape), 1: AAC-ELD */
  const int32_t *spec;          /* [n_ch][frame_length] (not modified) */
  const uint8_t *window_shape;  /* [n_ch] ics->window_shape of this frame */
  int32_t *overlap;             /* in/out: LD [n_ch][frame_length / 2], ELD [n_ch][3 * frame_length] (ptr_overlap_buf; zero for a
                                   new stream; ELD never writes its last frame_length / 4 words) */
  uint8_t *shape_prev;          /* [n_ch] in/out: ia_aac_dec_overlap_info.window_shape */
  int16_t *pcm16;               /* [n_ch * frame_length], channel-frame i at (i / ch_fac) * frame_length * ch_fac + i % ch_fac, stride ch_fac */
  int32_t *status;              /* [n_ch] or NULL: XAAC_OK, or XAAC_FATAL_BAD_WINDOW_SEQ for a shape byte > 1 (left untouched) */
} xaac_imdct_ld_batch;

/* ---- SBR QMF banks (fixed-point "Path B") ---------------------------------------------
 * xaac_qmf_analysis_batch  <-> ixheaacd_cplx_anal_qmffilt
 *      decl decoder/ixheaacd_qmf_dec.h:74, def decoder/generic/ixheaacd_qmf_dec_generic.c:590,
 *      call site decoder/ixheaacd_sbr_dec.c:1025
 * xaac_qmf_synthesis_batch <-> ixheaacd_cplx_synt_qmffilt (no-PS branch)
 *      def decoder/ixheaacd_qmf_dec.c:811, call site decoder/ixheaacd_sbr_dec.c:1273
 * One frame (32 QMF slots) per channel per call, N channels per batch.  The persistent
 * state is the reference's own (ia_sbr_qmf_filter_bank_struct, decoder/ixheaacd_qmf_dec.h:23-72)
 * with pointers turned into offsets, so it can be copied to and from a reference decoder. */
typedef struct xaac_qmf_ana_state {
  int16_t ring[320]; /* anal_filter_states */
  int16_t wr;        /* core_samples_buffer - anal_filter_states */
  int16_t phase;     /* filter_pos - qmf_c */
} xaac_qmf_ana_state;  /* zero-initialised for a new stream (sbrdec_initfuncs.c:1105-1122) */

/* The LD / ELD flavour of the complex analysis bank (ixheaacd_cplx_anal_qmffilt with AOT_ER_AAC_LD / _ELD,
 * generic/ixheaacd_qmf_dec_generic.c:590-741: low-delay prototype qmf_c_eld3, ixheaacd_sbr_qmfanal32_winadd_eld
 * qmf_dec.c:484-535, ELD post-modulation twiddles): the reference keeps four rotating pointers between frames. */
typedef struct xaac_qmf_ana_eld_state {
  int16_t ring[320]; /* anal_filter_states */
  int16_t wr;        /* core_samples_buffer - anal_filter_states */
  int16_t f1;        /* filter_pos - qmf_c_eld3 */
  int16_t f2;        /* filter_2 - qmf_c_eld3 */
  int16_t fp;        /* fp1_anal - anal_filter_states (0 or 32; fp2_anal is the other) */
} xaac_qmf_ana_eld_state; /* a new stream: ring zero, {0, 0, 32, 0} (sbrdec_initfuncs.c:1122-1148) */

typedef struct xaac_qmf_ana_eld_batch {
  int32_t n_ch;
  int32_t n_slots;           /* QMF slots per frame: 16 (512-sample frames) or 15 (480) */
  int32_t usb;               /* bands that get the post-modulation rotation (qmf_bank->usb; lsb is 0) */
  int32_t slot_stride;       /* words between consecutive slots in qmf (>= 96: imaginary bands at +64) */
  const int16_t *pcm;        /* [n_ch][32 * n_slots] core-decoder PCM16, planar */
  xaac_qmf_ana_eld_state *state; /* [n_ch] in/out */
  int32_t *qmf;              /* [n_ch][n_slots][slot_stride]: 32 real bands at +0, 32 imaginary at +64 */
  int32_t *status;           /* [n_ch] or NULL: 0, or -1 for a state whose four pointers are not one of the ten phases the
                                bank can be in (such a channel is left alone) */
} xaac_qmf_ana_eld_batch;

/* ... and of the complex synthesis bank (ixheaacd_cplx_synt_qmffilt with AOT_ER_AAC_LD / _ELD, decoder/ixheaacd_qmf_dec.c:
 * 811-1135: pre-twiddle :788, 64-channel inverse modulation, ixheaacd_shiftrountine_with_rnd_eld generic:1672, the
 * 10-tap window-add on qmf_c_eld with output shift 2), no parametric stereo, no DRC */
typedef struct xaac_qmf_syn_eld_state {
  int16_t ring[1280]; /* filter_states */
  int16_t drc_offset; /* ixheaacd_drc_offset */
  int16_t phase;      /* filter_pos_syn - qmf_c_eld */
  int16_t fp;         /* fp1_syn - filter_states (0 or 64) */
  int16_t sixty4;     /* sixty4 (64 or -64; fp2_syn = fp1_syn + sixty4) */
} xaac_qmf_syn_eld_state; /* a new stream: ring zero, {0, 0, 0, 64} (sbrdec_initfuncs.c:1181-1209) */

typedef struct xaac_qmf_syn_eld_batch {
  int32_t n_ch;
  int32_t n_slots;           /* 16 or 15 */
  int32_t lsb, usb, split;   /* region rescale as in xaac_qmf_syn_batch (qmf_dec.c:937-953) */
  int32_t slot_stride;       /* >= 128: imaginary row at +64 */
  const int32_t *qmf;        /* [n_ch][n_slots][slot_stride] (not modified) */
  const int16_t *scale;      /* [n_ch][4]: lb_scale, ov_lb_scale, hb_scale, st_syn_scale */
  xaac_qmf_syn_eld_state *state; /* [n_ch] in/out */
  int16_t *pcm;              /* [n_ch][64 * n_slots] planar */
  int32_t *status;           /* [n_ch] or NULL: -1 for a state outside the bank's ten phases (left alone) */
  int32_t *qmf_scaled;       /* optional [n_ch][n_slots][slot_stride], words 0..127 of a slot: the region-rescaled matrix, which the
                                reference makes in place of its input and hands on through qmf_real_out / qmf_imag_out
                                (qmf_dec.c:937-976; its input rows are work space afterwards) */
} xaac_qmf_syn_eld_batch;

typedef struct xaac_qmf_syn_state {
  int16_t ring[1280]; /* filter_states */
  int16_t drc_offset; /* ixheaacd_drc_offset */
  int16_t phase;      /* filter_pos_syn - qmf_c */
} xaac_qmf_syn_state;  /* zero-initialised for a new stream (sbrdec_initfuncs.c:1175-1196) */

typedef struct xaac_qmf_ana_batch {
  int32_t n_ch;              /* channels in the batch */
  int32_t ch_fac;            /* interleave stride of pcm: channel i reads sample n at
                                (i/ch_fac)*1024*ch_fac + n*ch_fac + i%ch_fac */
  int32_t low_pow;           /* 1: real-valued low-power bank (dct3_32), 0: complex HQ bank */
  int32_t usb;               /* analysis bank usb (HQ: bands that get the final rotation) */
  int32_t slot_stride;       /* words between consecutive slots in qmf (>= 32, >= 96 for HQ) */
  const int16_t *pcm;        /* core-decoder PCM16, 1024 samples per channel */
  xaac_qmf_ana_state *state; /* [n_ch] in/out */
  int32_t *qmf;              /* [n_ch][32][slot_stride]: 32 real bands at +0, 32 imaginary at +64 (HQ) */
} xaac_qmf_ana_batch;

typedef struct xaac_qmf_syn_batch {
  int32_t n_ch;
  int32_t ch_fac;            /* interleave stride of the 2048-sample PCM output */
  int32_t low_pow;
  int32_t lsb, usb;          /* synthesis bank lsb / usb (region rescale, qmf_dec.c:937-953) */
  int32_t split;             /* first slot of the current frame's low band (op_delay = 6) */
  int32_t slot_stride;       /* >= 64 (LP) / >= 128 (HQ: imaginary row at +64) */
  int32_t down_sample;       /* 1: the down-sampled bank (32 channels, sbrdec_initfuncs.c:1165; -dsample or output rates
                                above 48 kHz): bands 0..31 only, 1024 samples out, 640 ring samples in use */
  const int32_t *qmf;        /* [n_ch][32][slot_stride] (not modified) */
  const int16_t *scale;      /* [n_ch][4]: lb_scale, ov_lb_scale, hb_scale, st_syn_scale */
  xaac_qmf_syn_state *state; /* [n_ch] in/out */
  int16_t *pcm;              /* 2048 (down_sample: 1024) samples per channel, interleaved at ch_fac */
} xaac_qmf_syn_batch;

/* ---- low-power SBR, whole channel-frame (HE-AACv1) ---------------------------------------
 * xaac_sbr_lp_process_batch <-> ixheaacd_sbr_dec with low_pow_flag = 1
 *      def decoder/ixheaacd_sbr_dec.c:662, call sites decoder/ixheaacd_sbrdecoder.c:877 / :943
 * N independent channels, one frame each: core PCM16 (1024) + SBR side info -> 2048 PCM16, with the
 * per-channel state of xaac_sbr_state carried in device memory.  Formats: xaac_sbr.h. */
typedef struct xaac_sbr_lp_batch {
  int32_t n_ch;
  int32_t in_ch_fac, out_ch_fac;   /* interleave strides of pcm_in (1024/ch) and pcm_out (2048/ch) */
  int32_t down_sample;             /* 1: down-sampled synthesis bank, 1024 samples out per channel */
  const int16_t *pcm_in;
  const xaac_sbr_header *header;   /* [n_ch] (channels of one stream carry copies) */
  const xaac_sbr_frame *frame;     /* [n_ch] */
  xaac_sbr_state *state;           /* [n_ch] in/out */
  int16_t *pcm_out;
  int32_t *status;                 /* optional [n_ch]: 0, or -1 where the reference would have failed the frame */
  void *workspace;                 /* device scratch, >= xaac_sbr_lp_workspace_bytes(n_ch) */
  uint64_t workspace_bytes;
} xaac_sbr_lp_batch;

/* --- HQ (complex) SBR channel-frames, optionally with parametric stereo ---------------------------
 * ref: the same ixheaacd_sbr_dec with low_pow_flag = 0 -- HE-AAC mono and, with PS side info, HE-AACv2
 *      (decoder/ixheaacd_sbr_dec.c:1246-1281: the left channel's synthesis bank runs the PS tool, the
 *      right channel's bank consumes what it leaves).
 * N independent streams, one frame each: mono core PCM16 (1024) + SBR (+ PS) side info -> 2048 PCM16, or
 * with PS 2048 L,R pairs.  With PS every stream's channel_mode must be PS_STEREO; a stream whose frame
 * is not processed (apply_processing = 0) gets its mono output on the left and its right-channel
 * buffer and bank untouched, as in the reference. */
typedef struct xaac_sbr_hq_batch {
  int32_t n_ch;                    /* streams */
  int32_t in_ch_fac;               /* interleave stride of pcm_in (1024 per stream) */
  int32_t out_ch_fac;              /* without PS: interleave stride of pcm_out; with PS the output is L,R pairs */
  int32_t down_sample;             /* 1: down-sampled synthesis bank (1024 samples out); not together with PS -- the
                                      reference itself hands the right bank half a slot there (qmf_dec.c:1117-1119) */
  const int16_t *pcm_in;
  const xaac_sbr_header *header;   /* [n_ch] */
  const xaac_sbr_frame *frame;     /* [n_ch] */
  xaac_sbr_state *state;           /* [n_ch] in/out */
  const xaac_ps_frame *ps_frame;   /* [n_ch], or NULL together with ps_state: no parametric stereo */
  xaac_ps_state *ps_state;         /* [n_ch] in/out */
  int16_t *pcm_out;                /* n_ch x 2048 (x 2 with PS) */
  int32_t *status;                 /* optional [n_ch] */
  void *workspace;                 /* device scratch, >= xaac_sbr_hq_workspace_bytes(n_ch, ps_frame != NULL) */
  uint64_t workspace_bytes;
  int32_t max_band_hint;           /* optional hint, like xaac_esbr_sbr_batch.hbe_max_synth_size: 48 = the caller knows that no stream of
                                      the batch reaches above QMF band 48 -- SBR range, frequency tables, patches, the banks' band limits of
                                      this frame and of the frames still in its overlap rows (true once a stream's headers have kept
                                      sub_band_end <= 48 since its start, as the 24 / 32 kHz-core streams of HE-AACv2 services do).  The core
                                      then works on 48-band rows only and the launch that takes other streams through 64-band rows is
                                      left out; a stream for which the assertion does not hold is refused (status XAAC_FATAL_BAD_ARG:
                                      the frame is not decoded, the stream cannot be continued from its state).  0: no assertion, any stream
                                      is decoded. */
} xaac_sbr_hq_batch;

/* --- low-delay SBR channel-frames of AAC-ELD ---------------------------------------------------------------------------
 * ref: ixheaacd_sbr_dec with audio_object_type AOT_ER_AAC_ELD, low_pow_flag 0 (decoder/ixheaacd_sbr_dec.c:706-775, :1025-1308):
 *      the LD complex analysis bank (generic:590), block floating point, ixheaacd_hf_generator and ixheaacd_calc_sbrenvelope
 *      on a frame of 16 (512-sample core frames) or 15 (480) QMF slots with one slot per time slot and no overlap slots
 *      (op_delay 0: nothing of the matrix is carried; lpp_tran.c:1027-1052, env_calc.c:811-849, :962), the LD complex
 *      synthesis bank (qmf_dec.c:811).  One channel per entry: core PCM16 32 n_slots samples in, 64 n_slots out. */
typedef struct xaac_sbr_eld_state {
  xaac_qmf_ana_eld_state ana;                 /* str_codec_qmf_bank */
  xaac_qmf_syn_eld_state syn;                 /* str_synthesis_qmf_bank */
  int16_t codec_usb;                          /* str_codec_qmf_bank.usb */
  int16_t syn_lsb, syn_usb;                   /* str_synthesis_qmf_bank.lsb / .usb */
  int16_t pad2_;
  XAAC_SBR_STATE_TAIL_FIELDS
} xaac_sbr_eld_state; /* a new stream: the banks' initial states (above), xaac_sbr_state_init's values for the rest */

typedef struct xaac_sbr_eld_batch {
  int32_t n_ch;
  int32_t n_slots;                 /* 16 or 15: header num_time_slots (time_step 1, num_columns = n_slots) of every channel */
  int32_t in_ch_fac, out_ch_fac;   /* interleave strides of pcm_in (32 n_slots samples per channel) and pcm_out (64 n_slots) */
  const int16_t *pcm_in;
  const xaac_sbr_header *header;   /* [n_ch] */
  const xaac_sbr_frame *frame;     /* [n_ch] */
  xaac_sbr_eld_state *state;       /* [n_ch] in/out */
  int16_t *pcm_out;
  int32_t *status;                 /* optional [n_ch]: 0, -1 where the reference returns an error or the side info is outside the tables */
  void *workspace;                 /* device scratch, >= xaac_sbr_eld_workspace_bytes(n_ch) */
  uint64_t workspace_bytes;
  int32_t *qmf_handed_on;          /* optional [n_ch][n_slots][128]: the region-rescaled matrix the reference hands on through
                                      p_arr_qmf_buf_real / _imag (qmf_dec.c:966-976) */
} xaac_sbr_eld_batch;

/* ---- state hand-overs at a change of channel configuration ------------------------------------------------------
 * xaac_sbr_state_handover <-> the two memcpy blocks of ixheaacd_sbr_dec_apply, decoder/ixheaacd_sbrdecoder.c:762-806:
 * a stream that was mono (no PS) in the previous frame and carries parametric stereo now starts the right synthesis
 * bank from the left one's filter states (:762-775); a stream that turns from mono to stereo starts channel 1 from
 * channel 0's synthesis / analysis filter states, overlap buffer and their scale factors (:777-806).  The host calls it
 * before the frame's xaac_sbr_*_process_batch for exactly the streams whose configuration changed. */
#define XAAC_HANDOVER_PS_START 1
#define XAAC_HANDOVER_STEREO_START 2
typedef struct xaac_sbr_handover_batch {
  int32_t n;                /* entries */
  int32_t mode;             /* XAAC_HANDOVER_PS_START or XAAC_HANDOVER_STEREO_START */
  const int32_t *src;       /* [n] device: index into state of the channel that was running (channel 0) */
  const int32_t *dst;       /* [n] device: PS_START: index into ps_state; STEREO_START: index into state (channel 1) */
  xaac_sbr_state *state;
  xaac_ps_state *ps_state;  /* PS_START only */
} xaac_sbr_handover_batch;

/* ---- what a header change does to the resident state --------------------------------------------------------------
 * xaac_sbr_state_apply_side_batch <-> the words of the channel state that ixheaacd_sbr_dec_reset
 * (decoder/ixheaacd_sbrdecoder.c:103-252) and ixheaacd_prepare_upsamp (:254-276) rewrite, for every stream of the batch
 * whose frame carries the reset or the up-sampling flag -- on the device-resident arrays, so that a step in which every
 * stream resets (the first frame of a batch) costs one launch instead of a round trip of every state over the bus.
 * The same words as the host-side xaac_sbr_state_apply_side / xaac_ps_state_apply_side (include/xaac_parse.h) write.
 * Call it before the frame's xaac_sbr_*_process_batch, with that call's header array. */
typedef struct xaac_sbr_apply_side_batch {
  int32_t n_streams;
  int32_t ch_fac;                 /* channels per stream, 1 or 2: channel c of stream i is entry i * ch_fac + c */
  const xaac_sbr_header *header;  /* [n_streams * ch_fac] device: this frame's headers (sub_band_start / sub_band_end) */
  const int32_t *flags;           /* [n_streams][8] device: the parser's flag rows (xaac_parse_batch: [1] reset, [2]
                                     reset_channels, [3] upsampling); all zero for a stream without a frame */
  xaac_sbr_state *state;          /* [n_streams * ch_fac] device, in / out */
  xaac_ps_state *ps_state;        /* optional [n_streams] device (mono + PS streams: the right channel's bank), in / out */
} xaac_sbr_apply_side_batch;

/* ---- peak limiter + PCM16 hand-off (the AAC-LC post stage) ---------------------------------
 * xaac_peak_limiter_process_batch <-> ixheaacd_peak_limiter_process
 *      def decoder/ixheaacd_peak_limiter.c:201-309, call site decoder/ixheaacd_api.c:3667, followed by the
 *      round16 loop of api.c:3676-3681 (pcm16).  It consumes what xaac_imdct_process_batch leaves: the
 *      interleaved WORD32 block (out32) and qshift_adj per channel.
 * The state is ia_peak_limiter_struct (decoder/ixheaacd_peak_limiter_struct_def.h:31-51) with its two
 * buffer pointers turned into arrays: max_buf = attack_time_samples window of channel-maximum magnitudes,
 * delayed_input = the look-ahead delay line, attack_time_samples x num_channels interleaved. */
#define XAAC_LIM_MAX_ATTACK 480 /* 5 ms at 96 kHz */
#define XAAC_LIM_MAX_CH 8
typedef struct xaac_limiter_state {
  float attack_constant, release_constant;
  uint32_t num_channels;        /* 1 .. XAAC_LIM_MAX_CH */
  uint32_t attack_time_samples; /* 1 .. XAAC_LIM_MAX_ATTACK */
  uint32_t limiter_on;
  float gain_modified;
  float min_gain;               /* out: smallest gain applied in the last frame */
  uint32_t delayed_input_index;
  double pre_smoothed_gain;
  int32_t max_idx, cir_buf_pnt;
  float max_buf[XAAC_LIM_MAX_ATTACK];
  float delayed_input[XAAC_LIM_MAX_ATTACK * XAAC_LIM_MAX_CH];
} xaac_limiter_state;

typedef struct xaac_limiter_batch {
  int32_t n_streams;
  int32_t frame_len;          /* samples per channel, 1 .. 1024 */
  int32_t *samples;           /* in/out [n_streams][frame_len][num_channels] WORD32 (stream s at s * stride) */
  int64_t stride;             /* words between consecutive streams' blocks, >= frame_len * num_channels */
  const int8_t *qshift_adj;   /* [n_streams][num_channels], each 0 .. 30 */
  xaac_limiter_state *state;  /* [n_streams] in/out; num_channels must be the same in the whole batch */
  int32_t num_channels;
  int32_t planar;             /* 1: samples is [n_streams][num_channels][frame_len] -- what xaac_imdct_process_batch writes
                                 with ch_fac = 1 (16-byte stores) -- and stays so in place; pcm16 is interleaved either way */
  int16_t *pcm16;             /* optional [n_streams][frame_len][num_channels]: round16 of the result (dense) */
  int32_t *status;            /* optional [n_streams]: 0, or -1 for a stream whose state does not fit the batch
                                 (num_channels / attack_time_samples out of range): left untouched */
  void *workspace;            /* device scratch, >= xaac_peak_limiter_workspace_bytes(n_streams) */
  uint64_t workspace_bytes;
} xaac_limiter_batch;

/* ---- eSBR ("Path A", the reference's default -esbr:1) QMF banks ---------------------------------------------------
 * xaac_esbr_qmf_analysis_batch  <-> ixheaacd_esbr_analysis_filt_block
 *      def decoder/ixheaacd_sbr_dec.c:185 (32 analysis channels: HE-AAC 2:1), call site sbr_dec.c:892
 * xaac_esbr_qmf_synthesis_batch <-> the synthesis bank loop of ixheaacd_esbr_synthesis_filt_block
 *      def decoder/ixheaacd_sbr_dec.c:447 (64 synthesis channels, :572-656: after regrouping / PS / DRC factors)
 * Float at the edges, WORD32 inside (rings, 32-bit prototype filter and twiddles, 64-bit accumulation); every
 * conversion is exact, so the results are bit-identical to the reference's.  The states are the reference's WORD32
 * rings (ia_sbr_qmf_filter_bank_struct: anal_filter_states_32 / filter_states_32, decoder/ixheaacd_qmf_dec.h:54-58) with
 * their pointers as offsets. */
typedef struct xaac_esbr_ana_state {
  int32_t ring[320];   /* anal_filter_states_32 */
  int32_t pos;         /* state_new_samples_pos_low_32 - anal_filter_states_32 */
  int32_t win_off;     /* filter_pos_32 - esbr_qmf_c */
} xaac_esbr_ana_state; /* zero-initialised for a new stream (sbrdec_initfuncs.c:1108-1119) */

typedef struct xaac_esbr_syn_state {
  int32_t ring[1280];  /* filter_states_32 */
  int32_t drc_offset;  /* ixheaacd_drc_offset */
  int32_t filt_off;    /* filter_pos_syn_32 - esbr_qmf_c */
} xaac_esbr_syn_state; /* zero-initialised for a new stream (sbrdec_initfuncs.c:1185-1202) */

typedef struct xaac_esbr_ana_batch {
  int32_t n_ch;
  const float *core;           /* [n_ch][1024] core-decoder time samples (time_sample_buf) */
  xaac_esbr_ana_state *state;  /* [n_ch] in/out */
  float *qmf_re, *qmf_im;      /* [n_ch][32 slots][64]: bands 0..31 written (qmf_buf_real / _imag rows) */
} xaac_esbr_ana_batch;

/* The analysis banks of USAC's other SBR ratios (sbr_dec.c:213-236, sbrdec_initfuncs.c:766-816): 24 channels for 8:3 SBR (768
 * core samples -> 32 slots), 16 channels for 4:1 SBR (1024 -> 64 slots).  Same state struct (the ring holds 10 n_bands words;
 * win_off counts from esbr_qmf_c_24 for 24 channels). */
typedef struct xaac_esbr_ana_nb_batch {
  int32_t n_ch;
  int32_t n_bands;             /* 24 | 16 */
  int32_t n_slots;             /* <= 32 with 24 bands, <= 64 with 16 (n_bands * n_slots <= 1024) */
  int32_t core_stride;         /* floats between consecutive channels' core rows (>= n_bands * n_slots) */
  const float *core;           /* [n_ch][core_stride] */
  xaac_esbr_ana_state *state;  /* [n_ch] in/out */
  float *qmf_re, *qmf_im;      /* [n_ch][n_slots][64]: bands 0..n_bands-1 written, n_bands..31 zeroed */
} xaac_esbr_ana_nb_batch;

typedef struct xaac_esbr_syn_batch {
  int32_t n_ch;
  const float *qmf_re, *qmf_im; /* [n_ch][32 slots][64] */
  xaac_esbr_syn_state *state;   /* [n_ch] in/out */
  float *out;                   /* [n_ch][2048] time samples */
} xaac_esbr_syn_batch;

/* ---- USAC frequency-domain IMDCT ------------------------------------------------------------------------------------
 * xaac_usac_imdct_batch <-> ixheaacd_fd_frm_dec   def decoder/ixheaacd_imdct.c:596 (-> ixheaacd_fd_imdct_long :477,
 *      ixheaacd_fd_imdct_short :336, ixheaacd_acelp_imdct :186, ixheaacd_complex_fft_p2_dec ixheaacd_fft.c:1412),
 *      call site decoder/ixheaacd_ext_ch_ele.c:991, with the caller's float conversion and shape hand-over (:1008-1016).
 * Scope: ccfl = 1024 or 768 (the 768 / 96-line transforms run three power-of-two transforms and ixheaacd_complex_fft_p3's
 * three-point stage, ixheaacd_fft.c:2531); FD frames after FD frames and after LPD frames (td_frame_prev, FAC signal handed in:
 * see lpd_flags / fac below); no error concealment.  One frame of
 * one channel per entry; the overlap is the reference's overlap_data_ptr row (Q14, un-windowed), so a channel can move
 * between a reference decoder and this library at any frame boundary. */
typedef struct xaac_usac_ics {
  uint8_t window_sequence; /* 0 ONLY_LONG, 1 LONG_START, 2 EIGHT_SHORT, 3 LONG_STOP, 4 STOP_START (ixheaacd_cnst.h:100) */
  uint8_t window_shape;    /* 0 sine, 1 KBD */
} xaac_usac_ics;

typedef struct xaac_usac_fac {
  int32_t q;                 /* fac_q; a frame for which it would drive one of the windowing's shift counts outside 0..31
                                (given the frame's own transform exponent) is refused with XAAC_FATAL_BAD_ARG */
  int32_t data[256];         /* fac_idata[0 .. 2 lfac), lfac <= FAC_LENGTH = 128 */
} xaac_usac_fac;

typedef struct xaac_usac_fac_in {
  int32_t fac_data[129];     /* usac_data->fac_data[ch]: [0] the gain index, [1 .. lfac] the quantised FAC lines (not modified here;
                                the reference scales its copy in place, which nothing reads afterwards) */
  float lpc_prev[17];        /* usac_data->lpc_prev[ch]: the previous frame's LPC filter, ORDER + 1 coefficients */
  float acelp_in[256];       /* usac_data->acelp_in[ch][0 .. ccfl / 4): the ACELP zero-input response */
} xaac_usac_fac_in;

typedef struct xaac_usac_imdct_batch {
  int32_t n_ch;
  int32_t ccfl;              /* usac_data->ccfl: 1024 or 768; 0 = 1024.  (Added in round 3 where LP64 had four bytes of padding in
                                front of `coef`: the struct's size and the other members' offsets did not change, and a caller
                                built against the older header reads as ccfl 0 if it zero-initialised the descriptor, as the
                                samples in INTEGRATION.md do; one that did not has to be recompiled.  Members added since are
                                appended at the end of their structs.) */
  const int32_t *coef;       /* [n_ch][ccfl] coef_fix (not modified) */
  const xaac_usac_ics *ics;  /* [n_ch] */
  int32_t *overlap;          /* [n_ch][ccfl] in/out: overlap_data_ptr */
  uint8_t *shape_prev;       /* [n_ch] in/out: window_shape_prev */
  int32_t *out32;            /* optional [n_ch][ccfl]: output_data_ptr (Q15) */
  float *time;               /* optional [n_ch][ccfl]: time_sample_vector (= out32 * 2^-15) */
  int32_t *status;           /* optional [n_ch]: XAAC_OK or XAAC_FATAL_BAD_WINDOW_SEQ / _BAD_ARG (channel-frame left untouched) */
  /* LPD -> FD transitions (ixheaacd_fd_frm_dec :618-640).  lpd_flags, optional [n_ch]: bit 0 = usac_data->td_frame_prev
     (the slope behind an LPD frame is 2 lfac samples long, lfac = ccfl / 16 for EIGHT_SHORT, ccfl / 8 else), bit 1 =
     usac_data->fac_data_present.  fac, optional [n_ch]: the forward-aliasing-cancellation signal of the channels with bit
     1 set, as ixheaacd_cal_fac_data (:210) leaves it for the windowing -- fac_idata[0 .. 2 lfac) and its exponent: it is
     made from the LPD decoder's state (previous LPC filter, ACELP zero-input response), which stays on the host.
     Behind an LPD frame out32 / time are what the reference holds once ixheaacd_lpd_bpf_fix -- the LPD decoder's bass
     post filter, also LPD state -- has been the identity (Q15 -> float -> Q15 as imdct.c:459-470 does around it): the
     host applies its filter to `time` and converts back. */
  const uint8_t *lpd_flags;
  const struct xaac_usac_fac *fac;
  /* ... or the signal made on the device: fac_in, optional [n_ch] -- the LPD-side inputs of ixheaacd_cal_fac_data (imdct.c:210) for the
     channels with both flags set, as the LPD decoder leaves them in usac_data; the function then runs here (its transform of lfac
     lines, the previous LPC filter's weighted recursion, the zero-input response through the window's slopes) and `fac` is not
     read.  fac_work: [n_ch] scratch for the signals, required with fac_in.  A frame for which the reference's function returns an
     error is refused with XAAC_FATAL_BAD_ARG like one with an unusable exponent. */
  const struct xaac_usac_fac_in *fac_in;
  struct xaac_usac_fac *fac_work;
} xaac_usac_imdct_batch;

typedef struct xaac_ctx xaac_ctx;

/* Create a context bound to HIP device `device`.  `hip_stream` is a
 * hipStream_t to launch on (NULL: the context creates and owns one). */
XAAC_API int32_t xaac_create(xaac_ctx **ctx, int32_t device, void *hip_stream);
XAAC_API int32_t xaac_destroy(xaac_ctx *ctx);
/* Rebind the context to another hipStream_t; NULL here means the device's
 * legacy null stream (unlike xaac_create, nothing is created). */
XAAC_API int32_t xaac_set_stream(xaac_ctx *ctx, void *hip_stream);
/* Block until everything queued on the context's stream has finished. */
XAAC_API int32_t xaac_sync(xaac_ctx *ctx);
/* Loads every kernel's code object onto the context's device now (the HIP runtime otherwise does that at a module's first launch:
   some 20 ms inside the first batch a process decodes).  Optional; no reference counterpart. */
XAAC_API int32_t xaac_warm_up(xaac_ctx *ctx);

/* Enqueue one IMDCT + overlap-add pass over the batch (asynchronous). */
XAAC_API int32_t xaac_imdct_process_batch(xaac_ctx *ctx, const xaac_imdct_batch *batch);
/* Same with host buffers: copies in, runs, copies the outputs and state back,
 * synchronises.  PCIe-inclusive convenience path. */
XAAC_API int32_t xaac_imdct_process_batch_host(xaac_ctx *ctx, const xaac_imdct_batch *batch);
/* ixheaacd_imdct_process with ics->frame_length == 960 (the 960-line profile of DAB+ / DRM: ixheaacd_mdct_960 and eight
 * ixheaacd_inverse_transform_960, decoder/ixheaacd_aac_imdct.c:1672 / :1624, the 960- and 120-sample windows, the 960
 * branches of lpfuncs.c:347-802).  The same descriptor with 960 for 1024 and 480 for 512: spec [n_ch][960], overlap
 * [n_ch][480], out32 / pcm16 [n_ch * 960] interleaved at ch_fac; pcm16 is the plain hand-off of pcm_mode per sample. */
XAAC_API int32_t xaac_imdct960_process_batch(xaac_ctx *ctx, const xaac_imdct_batch *batch);
/* ixheaacd_imdct_process for AAC-LD / AAC-ELD frames (512 or 480 lines), device pointers, asynchronous */
XAAC_API int32_t xaac_imdct_ld_process_batch(xaac_ctx *ctx, const xaac_imdct_ld_batch *batch);

/* SBR QMF banks, device pointers, asynchronous on the context's stream. */
XAAC_API int32_t xaac_qmf_analysis_batch(xaac_ctx *ctx, const xaac_qmf_ana_batch *batch);
XAAC_API int32_t xaac_qmf_synthesis_batch(xaac_ctx *ctx, const xaac_qmf_syn_batch *batch);

/* USAC FD IMDCT + windowing + overlap-add, device pointers, asynchronous on the context's stream. */
XAAC_API int32_t xaac_usac_imdct_process_batch(xaac_ctx *ctx, const xaac_usac_imdct_batch *batch);

/* eSBR (Path A) QMF banks, device pointers, asynchronous on the context's stream. */
XAAC_API int32_t xaac_esbr_qmf_analysis_batch(xaac_ctx *ctx, const xaac_esbr_ana_batch *batch);
XAAC_API int32_t xaac_esbr_qmf_analysis_nb_batch(xaac_ctx *ctx, const xaac_esbr_ana_nb_batch *batch);
/* ixheaacd_cplx_anal_qmffilt for AAC-LD / ELD cores (complex bank, 16 or 15 slots per frame) */
XAAC_API int32_t xaac_qmf_analysis_eld_batch(xaac_ctx *ctx, const xaac_qmf_ana_eld_batch *batch);
/* ixheaacd_cplx_synt_qmffilt for AAC-LD / ELD (complex bank, 64 channels, 16 or 15 slots per frame) */
XAAC_API int32_t xaac_qmf_synthesis_eld_batch(xaac_ctx *ctx, const xaac_qmf_syn_eld_batch *batch);
XAAC_API int32_t xaac_esbr_qmf_synthesis_batch(xaac_ctx *ctx, const xaac_esbr_syn_batch *batch);
/* the same bank down-sampled (32 synthesis channels: -dsample, output rates above 48 kHz; sbr_dec.c:605-628): the same descriptor,
   out [n_ch][1024], the state's ring used up to word 640 */
XAAC_API int32_t xaac_esbr_qmf_synthesis_ds_batch(xaac_ctx *ctx, const xaac_esbr_syn_batch *batch);

/* Low-power SBR channel-frames (QMF analysis -> HF generation + envelope adjustment -> QMF synthesis). */
XAAC_API uint64_t xaac_sbr_lp_workspace_bytes(int32_t n_ch);
XAAC_API int32_t xaac_sbr_lp_process_batch(xaac_ctx *ctx, const xaac_sbr_lp_batch *batch);

/* HQ SBR stream-frames (complex QMF analysis -> LPP transposer + envelope adjustment -> [parametric
 * stereo] -> complex QMF synthesis, once per output channel). */
XAAC_API uint64_t xaac_sbr_eld_workspace_bytes(int32_t n_ch);
XAAC_API int32_t xaac_sbr_eld_process_batch(xaac_ctx *ctx, const xaac_sbr_eld_batch *batch);
XAAC_API uint64_t xaac_sbr_hq_workspace_bytes(int32_t n_ch, int32_t with_ps);
XAAC_API int32_t xaac_sbr_hq_process_batch(xaac_ctx *ctx, const xaac_sbr_hq_batch *batch);

/* Channel-configuration hand-overs (device pointers, asynchronous). */
XAAC_API int32_t xaac_sbr_state_handover(xaac_ctx *ctx, const xaac_sbr_handover_batch *batch);
XAAC_API int32_t xaac_sbr_state_apply_side_batch(xaac_ctx *ctx, const xaac_sbr_apply_side_batch *batch);

/* ixheaacd_peak_limiter_init (peak_limiter.c:46-77) on a host-side state; returns the limiter delay in
 * samples (attack_time_samples) or a fatal code when the rate / channel count does not fit the struct. */
XAAC_API int32_t xaac_peak_limiter_init(xaac_limiter_state *state, uint32_t num_channels, uint32_t sample_rate);
/* One frame of every stream through the limiter (device pointers, asynchronous). */
XAAC_API uint64_t xaac_peak_limiter_workspace_bytes(int32_t n_streams);
XAAC_API int32_t xaac_peak_limiter_process_batch(xaac_ctx *ctx, const xaac_limiter_batch *batch);

/* Launch geometry the library used for the last batch (for reports). */
XAAC_API int32_t xaac_last_launch(xaac_ctx *ctx, int32_t *grid, int32_t *block, int32_t *lds_bytes);
/* "libxaac_amd <version> gfx950" */
XAAC_API const char *xaac_version(void);

#ifdef __cplusplus
}
#endif
#endif /* XAAC_AMD_H */
