/*
 * oracle/ref_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin C entry points around the *real* reference functions so that Python
 * (ctypes) can drive them without mirroring the reference's structs.  Compiled
 * against the reference headers where they lie under /root/reference and linked
 * with the reference objects into oracle/_ref/libref_harness.so by
 * oracle/Makefile.ref.  Contains no reference code: it only fills the reference's
 * own structs (as decoder/ixheaacd_aacdecoder.c:192-209 does for the window
 * pointers) and calls the reference's external symbols.
 */
#include <stdio.h>
#include <string.h>
#include "ixheaacd_sbr_common.h"
#include "ixheaac_type_def.h"
#include "ixheaac_constants.h"
#include "ixheaac_basic_ops32.h"
#include "ixheaac_basic_ops16.h"
#include "ixheaac_basic_ops40.h"
#include "ixheaac_basic_ops.h"
#include "ixheaac_basic_op.h"
#include "ixheaacd_intrinsics.h"
#include "ixheaacd_common_rom.h"
#include "ixheaacd_sbrdecsettings.h"
#include "ixheaacd_bitbuffer.h"
#include "ixheaacd_defines.h"
#include "ixheaacd_pns.h"
#include "ixheaacd_aac_rom.h"
#include "ixheaacd_aac_imdct.h"
#include "ixheaacd_pulsedata.h"
#include "ixheaacd_drc_data_struct.h"
#include "ixheaacd_lt_predict.h"
#include "ixheaacd_cnst.h"
#include "ixheaacd_ec_defines.h"
#include "ixheaacd_ec_struct_def.h"
#include "ixheaacd_channelinfo.h"
#include "ixheaacd_drc_dec.h"
#include "ixheaacd_sbrdecoder.h"
#include "ixheaacd_tns.h"
#include "ixheaacd_sbr_scale.h"
#include "ixheaacd_lpp_tran.h"
#include "ixheaacd_env_extr_part.h"
#include "ixheaacd_sbr_rom.h"
#include "ixheaacd_block.h"
#include "ixheaacd_hybrid.h"
#include "ixheaacd_ps_dec.h"
#include "ixheaacd_env_extr.h"
#include "ixheaacd_basic_funcs.h"
#include "ixheaacd_env_calc.h"
#include "ixheaacd_audioobjtypes.h"

/* One channel, one frame through the reference's ixheaacd_imdct_process
 * (decoder/ixheaacd_lpfuncs.c:347).
 *   spec[1024]      in, clobbered (the reference works in place)
 *   overlap[512]    in/out  (ia_aac_dec_overlap_info.ptr_overlap_buf)
 *   prev_seq/shape  in/out  (overlap_info.window_sequence / window_shape)
 *   out[1024*ch_fac] written at stride ch_fac
 * returns ics.qshift_adj */
int ref_imdct_process(WORD32 *spec, WORD32 *overlap, WORD16 *prev_seq, WORD16 *prev_shape,
                      int seq, int shape, WORD32 *out, int ch_fac) {
  static __thread WORD32 scratch[2048];
  ia_aac_dec_overlap_info oi;
  ia_ics_info_struct ics;
  ia_aac_dec_tables_struct tabs;
  ia_aac_dec_imdct_tables_struct *rom = (ia_aac_dec_imdct_tables_struct *)&ixheaacd_imdct_tables;
  memset(&oi, 0, sizeof(oi));
  memset(&ics, 0, sizeof(ics));
  memset(&tabs, 0, sizeof(tabs));
  tabs.pstr_imdct_tables = rom;
  oi.ptr_long_window[0] = rom->only_long_window_sine;
  oi.ptr_short_window[0] = rom->only_short_window_sine;
  oi.ptr_long_window[1] = rom->only_long_window_kbd;
  oi.ptr_short_window[1] = rom->only_short_window_kbd;
  oi.window_shape = *prev_shape;
  oi.window_sequence = *prev_seq;
  oi.ptr_overlap_buf = overlap;
  ics.window_sequence = (WORD16)seq;
  ics.window_shape = (WORD16)shape;
  ics.frame_length = 1024;
  ixheaacd_imdct_process(&oi, spec, &ics, out, (WORD16)ch_fac, scratch, &tabs, AOT_AAC_LC, 0, 0);
  *prev_seq = oi.window_sequence;
  *prev_shape = oi.window_shape;
  return ics.qshift_adj;
}

/* n channel-frames in a C loop (for timing the reference as the CPU baseline):
 * per channel c: spec[c][1024] (clobbered), overlap[c][512], prev_seq/prev_shape[c],
 * seq/shape[c]; writes PCM16 at stride 1 per channel like ixheaacd_scale_adjust +
 * round16 would (AAC-LC, limiter off).  Thread-safe across disjoint channel ranges. */
void ref_imdct_batch(int n, WORD32 *spec, WORD32 *overlap, WORD16 *prev_seq, WORD16 *prev_shape,
                     const UWORD8 *seq, const UWORD8 *shape, WORD16 *pcm) {
  WORD32 out[1024];
  int c, i;
  for (c = 0; c < n; c++) {
    int q = ref_imdct_process(spec + 1024 * (size_t)c, overlap + 512 * (size_t)c, prev_seq + c, prev_shape + c,
                              seq[c], shape[c], out, 1);
    if (pcm)
      for (i = 0; i < 1024; i++) pcm[1024 * (size_t)c + i] = ixheaac_round16((WORD32)((UWORD32)out[i] << q));
  }
}
