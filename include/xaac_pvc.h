/*
 * xaac_pvc.h -- C ABI of the predictive vector coding (PVC) envelope decoder of the eSBR tools, batched.
 *
 * Replaces, at ixheaacd_sbr_dec's call site (decoder/ixheaacd_sbr_dec.c:931-953), for frames with sbr_mode == PVC_SBR:
 *   ixheaacd_qmf_enrg_calc (ixheaacd_sbr_dec.c:80-129): energies of the low band's QMF samples, pairs (2:1) or quads
 *     (4:1) of time slots averaged into 16 PVC time slots;
 *   ixheaacd_pvc_process (ixheaacd_pred_vec_block.c:176-240) with its four stages: grouping into three low sub-band
 *     groups in dB (:62), time smoothing over 16 / 4 / 12 / 3 slots (:109), prediction of the high groups' energies from
 *     the code book entry pvc_id selects (:131), expansion to QMF bands (:30); and the esg history shift (:234);
 *   the bookkeeping the call site does afterwards (sbr_dec.c:945, :951-953: prev_pvc_flg, prev_first_bnd_idx,
 *     prev_pvc_rate).
 * The output is pvc_dec_out_buf, the 16 x 64 float energies ixheaacd_sbr_env_calc takes as its PVC envelope.
 * Plain pointers and sizes; every pointer is DEVICE memory.  Bit-exact: the float words equal the reference's.
 */
#ifndef XAAC_PVC_H
#define XAAC_PVC_H

#include <stdint.h>

#include "xaac_amd.h"

#define XAAC_PVC_SLOTS 16 /* PVC_NUM_TIME_SLOTS */
#define XAAC_PVC_NB_LOW 3 /* PVC_NB_LOW */

/* what the payload parser and the call site hand over per frame (ia_pvc_data_struct members written by
   ixheaacd_read_esbr_pvc_envelope, env_extr.c:127; the call's scalars) */
typedef struct xaac_pvc_frame {
  uint8_t pvc_mode;             /* 1: eight high groups, 2: six */
  uint8_t ns_mode;              /* 1: the short smoothing window (4 / 3 slots) */
  uint8_t pvc_rate;             /* upsamp_fac: 2, or 4 (quads of time slots, 16 low bands) */
  uint8_t low_power;            /* low_pow_flag: energies from the real parts only */
  int16_t first_bnd_idx;        /* sub_band_start, 0 .. 32 */
  int16_t first_pvc_timeslot;   /* str_pvc_frame_info.border_vec[0], 0 .. 15 */
  uint16_t pvc_id[XAAC_PVC_SLOTS]; /* code book entry per PVC time slot, < 128 */
} xaac_pvc_frame;

/* what a channel carries between frames (the other members of ia_pvc_data_struct are per-call work values) */
typedef struct xaac_pvc_state {
  float esg[XAAC_PVC_SLOTS - 1][XAAC_PVC_NB_LOW]; /* esg rows 0 .. 14: the previous frames' grouped energies (dB) */
  int16_t prev_first_bnd_idx;
  uint16_t prev_pvc_id;
  uint8_t prev_pvc_flg;         /* the host clears it after a frame without PVC (sbr_dec.c:948) */
  uint8_t prev_pvc_rate;
  uint8_t reserved[2];
} xaac_pvc_state;

typedef struct xaac_pvc_batch {
  int32_t n_ch;
  const xaac_pvc_frame *frame;  /* [n_ch] */
  const float *qmf_re, *qmf_im; /* [n_ch][qmf_stride]: row SBR_HF_ADJ_OFFSET (= 2) of qmf_buf_real / _imag onwards, 64 floats per
                                   row; 32 rows (pvc_rate 2) or 64 rows (pvc_rate 4) are read, sub-bands 0..31 / 0..15 */
  int32_t qmf_stride;           /* floats from one channel's row 2 to the next channel's, >= 32 * 64 (the rows read must exist:
                                   32 of them for pvc_rate 2, 64 for pvc_rate 4: a pvc_rate 4 frame in a batch whose stride
                                   is below 64 * 64 is refused with status -1, and so is a pvc_rate 4 frame with
                                   first_bnd_idx > 16 -- the reference fills energies of sub-bands 0..15 only there) */
  xaac_pvc_state *state;        /* [n_ch] in/out */
  float *out;                   /* [n_ch][16][64]: pvc_dec_out_buf (sub-bands below first_bnd_idx: zero) */
  int32_t *status;              /* [n_ch] or NULL: 0, -1 = parameters outside the reference's tables (nothing written) */
} xaac_pvc_batch;

#ifdef __cplusplus
extern "C" {
#endif

XAAC_API int32_t xaac_pvc_process_batch(xaac_ctx *ctx, const xaac_pvc_batch *batch);

#ifdef __cplusplus
}
#endif

#endif /* XAAC_PVC_H */
