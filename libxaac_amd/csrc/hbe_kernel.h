/* hbe_kernel.h -- launch interface of the harmonic transposer's polyphase bank kernels (internal). */
#ifndef XAAC_HBE_KERNEL_H
#define XAAC_HBE_KERNEL_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/xaac_esbr.h"
#include "../../include/xaac_hbe.h"

/* the two polyphase banks (hbe_kernel.hip: xaac_hbe_banks_kernel) */
#define XAAC_HBE_BANKS_THREADS 256
#define XAAC_HBE_PHASE_SYNTH 1
#define XAAC_HBE_PHASE_ANAL 2
/* LDS floats of a channel of bank size s: the time signal of 32 + 1 columns, then the larger of the synthesis phase's (v of 9 + 32
   columns, the columns' inputs, transform work space: none for the size-20 matrix, 96 per column for the 24-point transforms) and
   the analysis phase's (u and results of 16 columns, work space: none / 192 per column for the 48-point transforms) */
#define XAAC_HBE_MAX2(a, b) ((a) > (b) ? (a) : (b))
#define XAAC_HBE_BANKS_LDS_FLOATS(s)                                                                                    \
  (33 * (s) + XAAC_HBE_MAX2(41 * 2 * (s) + 32 * (s) + ((s) == 20 ? 0 : ((s) == 12 ? 32 * 96 : 32 * 4 * (s))),          \
                            2 * 16 * 4 * (s) + ((s) == 20 ? 0 : ((s) == 12 ? 16 * 192 : 16 * 8 * (s)))))
/* ... of a launch whose largest bank is hint (4 .. 20; anything else: the largest need of all sizes, size 12's) */
#define XAAC_HBE_BANKS_LDS_FLOATS_FOR(hint)                                                                             \
  ((hint) == 4 ? XAAC_HBE_BANKS_LDS_FLOATS(4)                                                                           \
               : ((hint) == 8 ? XAAC_HBE_MAX2(XAAC_HBE_BANKS_LDS_FLOATS(8), XAAC_HBE_BANKS_LDS_FLOATS(4)) : XAAC_HBE_BANKS_LDS_FLOATS(12)))
#define XAAC_HBE_BANKS_LDS (XAAC_HBE_BANKS_LDS_FLOATS(12) * 4)
/* a channel of bank size s fits the LDS a launch with this hint takes (the banks kernel, the products kernel and the core
   kernel of the chain all leave a channel that does not alone: status -1) */
#define XAAC_HBE_LDS_OK(s, hint) (XAAC_HBE_BANKS_LDS_FLOATS(s) <= XAAC_HBE_BANKS_LDS_FLOATS_FOR(hint))

typedef struct XaacHbeBanksParams {
  int32_t n_ch, num_columns;    /* num_columns: of the synthesis phase (32 in the apply chain) */
  const float *qmf_re, *qmf_im; /* [n_ch][num_columns][64] (synthesis phase) */
  xaac_hbe_state *state;        /* [n_ch] */
  int32_t *status;              /* [n_ch] or NULL */
  const int32_t *pitch;         /* apply mode: [n_ch] or NULL */
  int32_t apply;                /* 1: as the first steps of ixheaacd_qmf_hbe_apply (time-signal shift, the re-initialisation
                                   while fft_ready is 0, the frame's parameter check, qmf_in_buf rows moved down) */
  /* inside the Path A chain (xaac_esbr_sbr_process_batch): the pitch comes from the side info and a channel whose
     frame has no SBR processing is skipped (sbr_dec.c:882); NULL elsewhere */
  const xaac_sbr_frame *frame;
  const xaac_esbr_side *side;
  int32_t in_stride;            /* floats between consecutive channels' qmf rows (2048 unless the chain hands in its own) */
  int32_t phases;               /* XAAC_HBE_PHASE_SYNTH | XAAC_HBE_PHASE_ANAL */
  int32_t lds_synth_size;       /* 4 or 8: no channel of the batch has a larger bank (less LDS per workgroup, more of them per
                                   CU; a channel that does is refused); anything else: sized for every bank */
} XaacHbeBanksParams;

#define XAAC_HBE_POST_THREADS 256
#define XAAC_HBE_POST_BLK_ROWS (17 * 15 + 16) /* block (band bt, column i) in row 17 bt + i */
#define XAAC_HBE_POST_LDS (XAAC_HBE_POST_BLK_ROWS * 25 * 4 + 5 * 16 * 32 * 8) /* the blocks (+ cross terms) of 16 bands x 16 columns; five planes of normalised samples */
typedef struct XaacHbePostParams {
  int32_t n_ch;
  xaac_hbe_state *state;
  const int32_t *pitch;
  float *pv_re, *pv_im;         /* [n_ch][32][64] */
  const xaac_sbr_frame *frame;  /* as in XaacHbeBanksParams */
  const xaac_esbr_side *side;
  int32_t pv_stride;            /* floats between consecutive channels' output rows (2048 standalone) */
  int32_t zero_outside;         /* 1: bands outside start_band .. end_band - 1 of the 32 rows are written as zeros (the chain's
                                   scratch rows; the reference leaves whatever its buffer held) */
  int32_t lds_synth_size;       /* as XaacHbeBanksParams: a channel the banks kernel refused for its size is left alone here too */
} XaacHbePostParams;

#define XAAC_HBE_DFT_LDS (32 * 128 * 4) /* u of 32 columns */
typedef struct XaacHbeDftParams {
  int32_t n_ch, no_bins;
  const float *time_in;
  int32_t in_stride;
  const float *coef_re, *coef_im;
  const int32_t *cfg;
  xaac_hbe_dft_anal_state *state;
  float *qmf_re, *qmf_im;
  int32_t *status;
  int32_t state_stride; /* bytes between consecutive channels' states; 0: sizeof(xaac_hbe_dft_anal_state) */
  int32_t chain;        /* 1: behind xaac_hbe_dft_core_kernel -- a channel that kernel refused or skipped (*status_in != 0) is left
                           alone, status is not written */
  const int32_t *status_in; /* chain: the first channel's word */
  int32_t status_stride;    /* chain: bytes between the channels' words */
  int32_t qmf_stride;       /* floats between consecutive channels' rows; 0: (no_bins + 2) * 64 */
  int32_t max_rows;         /* rows a channel owns (0: no_bins + 2): what the reference's clears reach beyond them is left out */
  int32_t zero_below;       /* 1: sub-bands below a_start of rows 0 .. no_bins - 1 are written as zeros too (scratch rows that
                               start with anything; the reference's buffer holds zeros there) */
  const xaac_sbr_frame *frame; /* inside the Path A chain: a channel without SBR processing is skipped; NULL elsewhere */
} XaacHbeDftParams;

/* the DFT transposer up to its output signal (hbe_kernel.hip: xaac_hbe_dft_core_kernel; arithmetic: hbe_dft.h): 256 threads
   per channel-frame.  LDS: the input and output signals, the two transforms' twiddles, and the larger of the synthesis bank's
   work space (41 columns of 2 s values, 32 x s inputs, 32 x 96 transform words) and the hop's (spectrum, transposed spectrum,
   magnitude, phase, transform scratch). */
#define XAAC_HBE_DFT_CORE_THREADS 256
#define XAAC_HBE_DFT_CORE_U_FLOATS XAAC_HBE_MAX2(41 * 32 + 32 * 16 + 32 * 96, 1536 + 1540 + 772 + 772 + 768)
#define XAAC_HBE_DFT_CORE_LDS ((2 * XAAC_HBE_DFT_MAX_ANA + 4 * XAAC_HBE_DFT_MAX_SYN + 2 * 768 + XAAC_HBE_DFT_CORE_U_FLOATS) * 4)
typedef struct XaacHbeDftCoreParams {
  int32_t n_ch;
  const float *qmf_re, *qmf_im; /* [n_ch][32][64] */
  const int32_t *pitch, *oversampling, *cfg;
  const xaac_hbe_dft_cfg *cfg_tab;
  xaac_hbe_dft_state *state;
  int32_t *status;
  /* inside the Path A chain (xaac_esbr_sbr_process_batch): pitch and oversampling come from the side info, a channel whose frame
     has no SBR processing is skipped (sbr_dec.c:880; its last_status becomes 1: nothing ran); NULL elsewhere */
  const xaac_sbr_frame *frame;
  const xaac_esbr_side *side;
} XaacHbeDftCoreParams;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t xaac_launch_hbe_dft_anal(const XaacHbeDftParams *p, hipStream_t stream);
hipError_t xaac_launch_hbe_dft_core(const XaacHbeDftCoreParams *p, hipStream_t stream);
hipError_t xaac_launch_hbe_banks(const XaacHbeBanksParams *p, hipStream_t stream);
hipError_t xaac_launch_hbe_post(const XaacHbePostParams *p, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
