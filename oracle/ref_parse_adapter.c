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
