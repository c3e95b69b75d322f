/*
 * ref_pvc_adapter.c -- TEST INFRASTRUCTURE, part of oracle/_ref/libref_harness.so: calls the compiled reference's own
 * ixheaacd_qmf_enrg_calc (decoder/ixheaacd_sbr_dec.c:80) and ixheaacd_pvc_process (decoder/ixheaacd_pred_vec_block.c:176) the way
 * ixheaacd_sbr_dec does for a PVC frame (sbr_dec.c:704-705, :931-953), on the boundary structs of include/xaac_pvc.h.
 * Contains no reference code.
 */
#include "ref_convert.h"
#include "xaac_pvc.h"

VOID ixheaacd_qmf_enrg_calc(ia_sbr_dec_struct *ptr_sbr_dec, WORD32 upsample_ratio_idx, WORD32 low_pow_flag);

static ia_sbr_dec_struct pvc_dec;

/* qmf_re / qmf_im: row 2 of the QMF buffers onwards (64 rows of 64).  Returns what ixheaacd_pvc_process returns. */
int ref_pvc_process(const xaac_pvc_frame *f, const float *qmf_re, const float *qmf_im, xaac_pvc_state *st, float *out) {
  ia_pvc_data_struct pd;
  int i, rc;
  memset(&pd, 0, sizeof(pd));
  pd.pvc_mode = f->pvc_mode;
  pd.ns_mode = f->ns_mode;
  pd.pvc_rate = f->pvc_rate; /* sbr_dec.c:931 */
  for (i = 0; i < 16; i++) pd.pvc_id[i] = f->pvc_id[i];
  memcpy(pd.esg, st->esg, sizeof(st->esg));
  pd.prev_first_bnd_idx = st->prev_first_bnd_idx;
  pd.prev_pvc_id = st->prev_pvc_id;
  pd.prev_pvc_flg = st->prev_pvc_flg;
  pd.prev_pvc_rate = st->prev_pvc_rate;
  memcpy(&pvc_dec.qmf_buf_real[2][0], qmf_re, sizeof(float) * 64 * 64);
  memcpy(&pvc_dec.qmf_buf_imag[2][0], qmf_im, sizeof(float) * 64 * 64);
  pvc_dec.str_codec_qmf_bank.num_time_slots = f->pvc_rate == 4 ? 64 : 32;
  memset(out, 0, sizeof(float) * 1024);                                   /* :704 */
  memset(pvc_dec.pvc_qmf_enrg_arr, 0, sizeof(pvc_dec.pvc_qmf_enrg_arr)); /* :705 */
  ixheaacd_qmf_enrg_calc(&pvc_dec, f->pvc_rate == 4 ? SBR_UPSAMPLE_IDX_4_1 : SBR_UPSAMPLE_IDX_2_1, f->low_power);
  rc = ixheaacd_pvc_process(&pd, f->first_bnd_idx, f->first_pvc_timeslot, pvc_dec.pvc_qmf_enrg_arr, out);
  if (rc) return rc;
  memcpy(st->esg, pd.esg, sizeof(st->esg));
  st->prev_pvc_id = pd.prev_pvc_id;
  st->prev_pvc_flg = 1;                          /* :945 */
  st->prev_first_bnd_idx = f->first_bnd_idx;     /* :951 */
  st->prev_pvc_rate = pd.pvc_rate;               /* :953 */
  return 0;
}
