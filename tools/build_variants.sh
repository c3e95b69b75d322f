#!/bin/bash
# Developer tool (run HERE, hipcc cross-compiles): build library variants with extra flags, one object directory each:
#   bash tools/build_variants.sh tagA "-DFOO=1" tagB "-DBAR" ...   -> libxaac_amd/libxaac_amd_<tag>.so
# then time them on the GPU box with tools/time_variants.sh tagA tagB ...
cd "$(dirname "$0")/../libxaac_amd/csrc"
while [ $# -ge 2 ]; do
  make -s OUT=$PWD/../libxaac_amd_$1.so OBJD=$PWD/build_$1 EXTRA="$2" 2>&1 | grep -E "error|Error" 
  shift 2
done
ls -la ../*.so
