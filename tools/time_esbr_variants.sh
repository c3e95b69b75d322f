#!/bin/bash
# Developer tool (GPU box): per-kernel times of bench.py's Path A leg (tools/prof_esbr.py) for prebuilt library variants
# (tools/build_variants.sh; "base" = the tree's libxaac_amd.so):  bash tools/time_esbr_variants.sh base prev ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for T in "$@"; do
  L=$R/libxaac_amd/libxaac_amd_$T.so
  [ "$T" = base ] && L=$R/libxaac_amd/libxaac_amd.so
  XAAC_AMD_LIBRARY=$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/e_$T -o r -- python $R/tools/prof_esbr.py > /tmp/e_$T.json 2>/dev/null
  echo "== $T: $(python -c "import json; d=json.loads(open('/tmp/e_$T.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['bit_exact_vs_oracle'], d['with_harmonic_transposer']['ms_per_step'])")"
  python $R/tools/rocprof_summary.py stats $(find /tmp/e_$T -name "*.db") | grep -E "esbr|hbe" | cut -c1-125
  rm -rf /tmp/e_$T
done
