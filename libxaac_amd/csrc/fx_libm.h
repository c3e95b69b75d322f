/*
 * fx_libm.h -- the four expressions in which the float tier (Path A: eSBR pre-flattening, the PVC envelope decoder)
 * goes through the C library's DOUBLE log10 / pow and rounds the result to float.  On the host (oracle, reference) that is
 * glibc; on the GPU it is ROCm's device library.  Neither is correctly rounded, so the two need not agree on every input --
 * they are held against each other over every float a call can receive by tests/test_libm_pin_gpu.py (tests/fuzz/
 * libm_probe.hip), which fails on a single differing float word outside the committed list of known inputs.
 * (cbrt, the third libm function on this tier, is restated in hbe_trans.h; sqrt is correctly rounded on both sides.)
 */
#ifndef XAAC_FX_LIBM_H
#define XAAC_FX_LIBM_H

#include <math.h>

#include "fx.h"

/* decoder/ixheaacd_pred_vec_block.c:74 (ixheaacd_pvc_qmf_grouping): (float)log10(esg), esg > 0.1 */
FX_HD float xm_log10f_of(float v) { return (float)log10((double)v); }
/* decoder/ixheaacd_pred_vec_block.c:46 (ixheaacd_pvc_sb_parsing): pow(10, r / 10) */
FX_HD float xm_pow10_tenth(float r) { return (float)pow(10.0, r / 10.0); }
/* decoder/ixheaacd_esbr_envcal.c (pre-flattening, ixheaacd_pre_processing): 10 * log10(t), t = mean energy + 1 */
FX_HD float xm_10log10f_of(float t) { return (float)(10 * log10((double)t)); }
/* ... and its gain curve: pow(10, a), a = (mean - slope) / 20 */
FX_HD float xm_pow10f_of(float a) { return (float)pow(10.0, (double)a); }

#endif /* XAAC_FX_LIBM_H */
