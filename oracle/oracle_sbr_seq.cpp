/*
 * oracle/oracle_sbr_seq.cpp -- TEST INFRASTRUCTURE ONLY: oracle_sbr.cpp compiled a second time with the envelope adjuster's
 * two-envelopes-per-pass arrangement switched off (XS_NO_ENV_PAIRS: every frame takes the one-envelope chain of
 * libxaac_amd/csrc/sbr_core.h, the literal restatement of ixheaacd_calc_sbrenvelope's loop, env_calc.c:692).  Exports
 * xo_sbr_dec_hq_seq; tests/test_env_pairs_cpu.py holds it against xo_sbr_dec_hq (pairs on) on the reference's captured
 * frames and on fuzzed side info.
 */
#define XS_NO_ENV_PAIRS 1
#define xo_sbr_dec_lp xo_sbr_dec_lp_seq
#define xo_sbr_dec_hq xo_sbr_dec_hq_seq
#define xo_sbr_dec_lp_batch xo_sbr_dec_lp_batch_seq
#define xo_sbr_dec_hq_batch xo_sbr_dec_hq_batch_seq
#define xo_sbr_dec_hq_phased xo_sbr_dec_hq_phased_seq
#define xo_sbr_dec_lp_ds xo_sbr_dec_lp_ds_seq
#define xo_sbr_dec_hq_ds xo_sbr_dec_hq_ds_seq
#define xo_sbr_dec_eld xo_sbr_dec_eld_seq
#include "oracle_sbr.cpp"
