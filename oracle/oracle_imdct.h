/* oracle/oracle_imdct.h -- TEST INFRASTRUCTURE ONLY; see oracle_imdct.c. */
#ifndef XAAC_ORACLE_IMDCT_H
#define XAAC_ORACLE_IMDCT_H
#include <stdint.h>

/* window_sequence values: decoder/ixheaacd_cnst.h:100-103 */
enum { XO_ONLY_LONG = 0, XO_LONG_START = 1, XO_EIGHT_SHORT = 2, XO_LONG_STOP = 3 };

#ifdef __cplusplus
extern "C" {
#endif
/* one channel, one 1024-sample frame; out written at stride s; returns qshift_adj */
int xo_imdct_process(const int32_t *spec, int32_t *ovl, int16_t *prev_seq, int16_t *prev_shape, int seq, int shape,
                     int32_t *out, int s);
void xo_pcm16(const int32_t *in, int in_stride, int16_t *pcm, int pcm_stride, int n, int qadj, int mode);
void xo_pcm16_block(int32_t *blk, const int8_t *qadj, int nch, int mode, int16_t *out);
void xo_imdct_batch(int nch, const int32_t *spec, int32_t *ovl, int16_t *prev_seq, int16_t *prev_shape,
                    const uint8_t *seq, const uint8_t *shape, int32_t *out32, int16_t *pcm, int8_t *qadj,
                    int pcm_mode);
#ifdef __cplusplus
}
#endif
#endif
