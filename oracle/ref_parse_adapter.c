/*
 * oracle/ref_parse_adapter.c -- TEST INFRASTRUCTURE ONLY.
 *
 * ROM access for tools/gen_tables_aac.py and tests/test_tables.py: the members of the reference's AAC block / Huffman ROM
 * structs (decoder/ixheaacd_aac_rom.h:25-110) by name, and the reference's own code word lookup
 * (ixheaacd_huff_sfb_table, decoder/ixheaacd_block.c:75; ixheaacd_huffman_decode, decoder/ixheaacd_env_extr.c:88) as a probe, so that the generator can list every code word
 * of every code book (code, length, index) without knowing how the reference packs its tables.  Contains no reference code.
 */
#include <stddef.h>
#include <string.h>
#include "ixheaac_type_def.h"
#include "ixheaacd_defines.h"
#include "ixheaacd_aac_rom.h"

VOID ixheaacd_huff_sfb_table(WORD32 it_bit_buff, WORD16 *huff_index, WORD32 *len, const UWORD16 *code_book_tbl,
                             const UWORD32 *idx_table);
VOID ixheaacd_huffman_decode(WORD32 it_bit_buff, WORD16 *h_index, WORD16 *len, const UWORD16 *input_table,
                             const UWORD32 *idx_table);

#define BLK(m) {#m, &ixheaacd_aac_block_tables.m, sizeof(ixheaacd_aac_block_tables.m), sizeof(ixheaacd_aac_block_tables.m[0])}
#define HUF(m) \
  {#m, &ixheaacd_aac_huffmann_tables.m, sizeof(ixheaacd_aac_huffmann_tables.m), sizeof(ixheaacd_aac_huffmann_tables.m[0])}

static const struct {
  const char *name;
  const void *p;
  int bytes, elem;
} k_rom[] = {
    BLK(ixheaacd_pow_table_Q13), BLK(scale_table), BLK(tns_max_bands_tbl), BLK(tns_coeff3_16), BLK(tns_coeff4_16),
    BLK(tns_coeff3), BLK(tns_coeff4), BLK(tns_coeff3_32), BLK(tns_coeff4_32), BLK(scale_mant_tab),
    HUF(ixheaacd_sfb_96_1024), HUF(ixheaacd_sfb_96_128), HUF(ixheaacd_sfb_64_1024), HUF(ixheaacd_sfb_48_1024),
    HUF(ixheaacd_sfb_48_128), HUF(ixheaacd_sfb_32_1024), HUF(ixheaacd_sfb_24_1024), HUF(ixheaacd_sfb_24_128),
    HUF(ixheaacd_sfb_16_1024), HUF(ixheaacd_sfb_16_128), HUF(ixheaacd_sfb_8_1024), HUF(ixheaacd_sfb_8_128),
    HUF(str_sample_rate_info),
};

/* the named ROM member: address, size in bytes, size of one element; NULL when there is no such member */
const void *ref_aac_rom(const char *name, int *bytes, int *elem) {
  for (unsigned i = 0; i < sizeof(k_rom) / sizeof(k_rom[0]); i++)
    if (!strcmp(name, k_rom[i].name)) {
      *bytes = k_rom[i].bytes;
      *elem = k_rom[i].elem;
      return k_rom[i].p;
    }
  return NULL;
}

/* one lookup in code book cb (1..11 spectral, 0 = the scale factor book) of the 32 bits in `word` (first bit = MSB);
   the reference's index of the code word and its length */
void ref_aac_huff_probe(int cb, unsigned word, int *index, int *len) {
  const ia_aac_dec_huffman_tables_struct *h = &ixheaacd_aac_huffmann_tables;
  const UWORD16 *book[12] = {h->huffman_code_book_scl, h->input_table_cb1, h->input_table_cb2, h->input_table_cb3,
                             h->input_table_cb4,       h->input_table_cb5, h->input_table_cb6, h->input_table_cb7,
                             h->input_table_cb8,       h->input_table_cb9, h->input_table_cb10, h->input_table_cb11};
  const UWORD32 *idx[12] = {h->huffman_code_book_scl_index, h->idx_table_hf1, h->idx_table_hf2, h->idx_table_hf3,
                            h->idx_table_hf4,               h->idx_table_hf5, h->idx_table_hf6, h->idx_table_hf7,
                            h->idx_table_hf8,               h->idx_table_hf9, h->idx_table_hf10, h->idx_table_hf11};
  WORD16 i = 0, l16 = 0;
  WORD32 l = 0;
  if (cb == 11) { /* the escape book goes through its own lookup (block.c:160, :346), the others through the one in
                     decoder/ixheaacd_env_extr.c:88 (block.c:215 of longblock.c, block.c:507 ...): the index words differ */
    ixheaacd_huff_sfb_table((WORD32)word, &i, &l, book[cb], idx[cb]);
  } else {
    ixheaacd_huffman_decode((WORD32)word, &i, &l16, book[cb], idx[cb]);
    l = l16;
  }
  *index = i;
  *len = l;
}

/* the reference's inverse quantiser of escape magnitudes (decoder/ixheaacd_channel.c:1055); returns its error code */
WORD32 ixheaacd_inv_quant(WORD32 *px_quant, WORD32 *ixheaacd_pow_table_Q13);
int ref_inv_quant(int q, int *out) {
  WORD32 v = q;
  const int rc = ixheaacd_inv_quant(&v, (WORD32 *)ixheaacd_aac_block_tables.ixheaacd_pow_table_Q13);
  *out = v;
  return rc;
}

/* ---- SBR / PS side info ROM (decoder/ixheaacd_sbr_rom.h:118-240, ixheaacd_common_rom.h:30) ---------------------------- */
#include "ixheaac_constants.h"
#include "ixheaacd_sbr_common.h"
#include "ixheaacd_bitbuffer.h"
#include "ixheaacd_sbrdecsettings.h"
#include "ixheaacd_common_rom.h"
#include "ixheaacd_sbr_scale.h"
#include "ixheaacd_lpp_tran.h"
#include "ixheaacd_env_extr_part.h"
#include "ixheaacd_sbr_rom.h"

typedef const UWORD16 *ia_huffman_data_type; /* decoder/ixheaacd_env_extr.h:36 */
WORD32 ixheaacd_ssc_huff_dec(ia_huffman_data_type t_huff, ia_bit_buf_struct *it_bit_buff);

/* one lookup in SBR code book `table` (order of k_sbr below): index and length */
void ref_sbr_huff_probe(int table, unsigned word, int *index, int *len) {
  const ia_env_extr_tables_struct *t = &ixheaacd_aac_dec_env_extr_tables;
  const WORD16 *inp[10] = {t->ixheaacd_t_huffman_env_1_5db_inp_table,     t->ixheaacd_f_huffman_env_1_5db_inp_table,
                           t->ixheaacd_t_huffman_env_3_0db_inp_table,     t->ixheaacd_f_huffman_env_3_0db_inp_table,
                           t->ixheaacd_t_huffman_env_bal_1_5db_inp_table, t->ixheaacd_f_huffman_env_bal_1_5db_inp_table,
                           t->ixheaacd_t_huffman_env_bal_3_0db_inp_table, t->ixheaacd_f_huffman_env_bal_3_0db_inp_table,
                           t->ixheaacd_t_huffman_noise_3_0db_inp_table,   t->ixheaacd_t_huffman_noise_bal_3_0db_inp_table};
  const WORD32 *idx[10] = {t->ixheaacd_t_huffman_env_1_5db_idx_table,     t->ixheaacd_f_huffman_env_1_5db_idx_table,
                           t->ixheaacd_t_huffman_env_3_0db_idx_table,     t->ixheaacd_f_huffman_env_3_0db_idx_table,
                           t->ixheaacd_t_huffman_env_bal_1_5db_idx_table, t->ixheaacd_f_huffman_env_bal_1_5db_idx_table,
                           t->ixheaacd_t_huffman_env_bal_3_0db_idx_table, t->ixheaacd_f_huffman_env_bal_3_0db_idx_table,
                           t->ixheaacd_t_huffman_noise_3_0db_idx_table,   t->ixheaacd_t_huffman_noise_bal_3_0db_idx_table};
  WORD16 i = 0, l = 0;
  ixheaacd_huffman_decode((WORD32)word, &i, &l, (const UWORD16 *)inp[table], (const UWORD32 *)idx[table]);
  *index = i;
  *len = l;
}

/* one value of PS code book `table` (0 iid_df, 1 iid_dt, 2 iid_df_fine, 3 iid_dt_fine, 4 icc_df, 5 icc_dt) read from the 32
   bits in `word` by the reference's own tree walk (env_extr.c:325): the value it returns and the bits it took */
void ref_ps_huff_probe(int table, unsigned word, int *value, int *len) {
  const ia_ps_tables_struct *t = &ixheaacd_aac_dec_ps_tables;
  const WORD16 *tab[6] = {t->huff_iid_df, t->huff_iid_dt, t->huff_iid_df_fine, t->huff_iid_dt_fine, t->huff_icc_df, t->huff_icc_dt};
  UWORD8 bytes[8] = {(UWORD8)(word >> 24), (UWORD8)(word >> 16), (UWORD8)(word >> 8), (UWORD8)word, 0, 0, 0, 0};
  ia_bit_buf_struct bb;
  memset(&bb, 0, sizeof(bb));
  ixheaacd_create_init_bit_buf(&bb, bytes, 8);
  const WORD32 before = bb.cnt_bits;
  *value = ixheaacd_ssc_huff_dec((ia_huffman_data_type)tab[table], &bb);
  *len = before - bb.cnt_bits;
}

/* the FIXFIX frame grids (sbr_frame_info1_2_4_16, used by env_extr.c:1763) flattened to 24 int16 per entry:
   frame_class, num_env, transient_env, num_noise_env, border_vec[9], freq_res[8], noise_border_vec[3]; and log2 table */
int ref_sbr_frame_info(int entry, short *out) {
  const ia_frame_info_struct *f = &ixheaacd_aac_dec_env_extr_tables.sbr_frame_info1_2_4_16[entry];
  int i, n = 0;
  if (entry < 0 || entry >= 7) return -1;
  out[n++] = f->frame_class, out[n++] = f->num_env, out[n++] = f->transient_env, out[n++] = f->num_noise_env;
  for (i = 0; i < MAX_ENVELOPES + 1; i++) out[n++] = f->border_vec[i];
  for (i = 0; i < MAX_ENVELOPES; i++) out[n++] = f->freq_res[i];
  for (i = 0; i < MAX_NOISE_ENVELOPES + 1; i++) out[n++] = f->noise_border_vec[i];
  return n;
}
const short *ref_log_dual_is_table(int *count) {
  *count = LOG_2_TABLE_SIZE;
  return ixheaacd_str_fft_n_transcendent_tables.log_dual_is_table;
}
