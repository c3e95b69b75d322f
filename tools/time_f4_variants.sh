#!/bin/bash
# Developer tool (GPU box): the f4 transforms' launch times (bench.py: secondary_f4 through tools/pmc_probe_f4.py) for
# prebuilt library variants (tools/build_variants.sh), under rocprofv3 --kernel-trace --stats.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for T in "$@"; do
  L=$R/libxaac_amd/libxaac_amd_$T.so
  [ "$T" = base ] && L=$R/libxaac_amd/libxaac_amd.so
  XAAC_AMD_LIBRARY=$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/f_$T -o r -- python $R/tools/pmc_probe_f4.py > /tmp/f_$T.txt 2>/dev/null
  echo "== $T"
  python $R/tools/rocprof_summary.py stats $(find /tmp/f_$T -name "*.db") | sed -n 2,12p | cut -c1-125
done
