/* oracle/oracle_qmf.h -- TEST INFRASTRUCTURE ONLY; see oracle_qmf.cpp. */
#ifndef XAAC_ORACLE_QMF_H
#define XAAC_ORACLE_QMF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* persistent state of the analysis bank: ia_sbr_qmf_filter_bank_struct.anal_filter_states[320],
   core_samples_buffer (as an index) and filter_pos (as an offset into qmf_c) */
typedef struct {
  int16_t ring[320];
  int16_t wr;
  int16_t phase;
} xo_qmf_ana_state;
/* synthesis bank: filter_states[1280], ixheaacd_drc_offset, filter_pos_syn (offset into qmf_c) */
typedef struct {
  int16_t ring[1280];
  int16_t drc_offset;
  int16_t phase;
} xo_qmf_syn_state;

void xo_radix4(const int16_t *w, int32_t *x, int index1, int index);
void xo_postradix4(int32_t *y, const int32_t *x);
void xo_postradix2(int32_t *y, const int32_t *x);
void xo_dct3_32(int32_t *in, int32_t *out);
void xo_cos_sin_mod(int32_t *s, int m);
void xo_fwd_modulation(const int32_t *in, int32_t *s, int nrot);
void xo_dct2_64_lp(int32_t *x, int16_t *b);
void xo_synth_hq_slot(int32_t *s, int16_t *b, int shift);
void xo_qmf_ana_init(xo_qmf_ana_state *st);
void xo_qmf_analysis(const int16_t *pcm, int stride, xo_qmf_ana_state *st, int low_pow, int usb, int32_t *qmf,
                     int slot_stride);
void xo_qmf_syn_init(xo_qmf_syn_state *st);
void xo_qmf_synthesis_slot_n(const int32_t *x, xo_qmf_syn_state *st, int slot, int low_pow, int out_scale, int16_t *pcm,
                             int stride, int ds);
void xo_qmf_synthesis_n(const int32_t *qmf, int slot_stride, const int16_t *sf, int lsb, int usb, int split,
                        xo_qmf_syn_state *st, int low_pow, int16_t *pcm, int stride, int ds);
void xo_qmf_synthesis_slot(const int32_t *x, xo_qmf_syn_state *st, int slot, int low_pow, int out_scale, int16_t *pcm,
                           int stride);
void xo_qmf_synthesis(const int32_t *qmf, int slot_stride, const int16_t *sf, int lsb, int usb, int split,
                      xo_qmf_syn_state *st, int low_pow, int16_t *pcm, int stride);
#ifdef __cplusplus
}
#endif
#endif
