#!/bin/bash
# GPU box: the committed streams under the reference decoder's flags, the unmodified decoder (twice: is it deterministic?) against
# oracle/_ref/xaacdec_dropin (its seams served by libxaac_amd.so): same bytes?  which calls ran where?   -> profiles/r06_*_flag_survey.txt
cd $GRAFT_REPO_ROOT
R=oracle/_ref
printf "%-22s %-22s %-5s %-6s %s\n" stream flags same ref2x "calls on the GPU (imdct sbr | 960 ld | eld-sbr | esbr [ds] | limiter) / sbr_dec left to the reference"
for f in tests/golden/streams/mix_aot2_64k.aac tests/golden/streams/mix_aot5_48k.aac tests/golden/streams/mix_aot29_32k.aac tests/golden/streams/mono_aot5_32k.aac \
         tests/golden/streams/harm_aot5_48k.aac tests/golden/streams/he_aot5_44k.aac tests/golden/streams_ld/lc960.aac tests/golden/streams_ld/ld512.aac \
         tests/golden/streams_ld/eld512.aac tests/golden/streams_ld/eld480.aac tests/golden/streams_usac/u21.aac tests/golden/streams_usac/u21sw.aac \
         tests/golden/streams_usac/u83.aac tests/golden/streams_usac/m41swpvc.aac tests/golden/streams_wide/mc6_aot5.aac tests/golden/streams_wide/he960_aot29.aac; do
  ex=""; m=${f%.aac}.txt; [ -f $m ] && ex="-mp4:1 -imeta:$m"
  for fl in "" "-esbr:0" "-dsample:1" "-dsample:1 -esbr:0" "-esbr_hq:1" "-esbr_ps:1" "-pcmsz:24" "-peak_limiter_off:1" "-downmix:1"; do
    $R/xaacdec -ifile:$f -ofile:/tmp/r1.wav $fl $ex > /dev/null 2>&1
    $R/xaacdec -ifile:$f -ofile:/tmp/r2.wav $fl $ex > /dev/null 2>&1
    $R/xaacdec_dropin -ifile:$f -ofile:/tmp/g.wav $fl $ex 2> /tmp/g.err > /dev/null
    g() { grep -oE "$1" /tmp/g.err | head -1 | grep -oE "^[0-9]+"; }
    det=$(cmp -s /tmp/r1.wav /tmp/r2.wav && echo yes || echo NO)
    same=$(cmp -s /tmp/r1.wav /tmp/g.wav && echo yes || echo NO)
    [ -s /tmp/r1.wav ] || same="-"
    printf "%-22s %-22s %-5s %-6s %s %s | %s %s | %s | %s [%s] | %s / %s\n" "$(basename $f .aac)" "[$fl]" $same $det \
      "$(g '[0-9]+ imdct_process and')" "$(grep -oE 'and [0-9]+ sbr_dec calls ran' /tmp/g.err | grep -oE '[0-9]+')" \
      "$(g '[0-9]+ imdct_process calls of 960')" "$(grep -oE 'and [0-9]+ of AAC-LD' /tmp/g.err | grep -oE '[0-9]+')" "$(g '[0-9]+ whole low-delay')" \
      "$(g '[0-9]+ sbr_dec calls took')" "$(g '[0-9]+ of the eSBR calls with the down')" "$(g '[0-9]+ peak_limiter')" "$(grep -oE '[0-9]+ sbr_dec calls left' /tmp/g.err | grep -oE '^[0-9]+')"
  done
done
