#!/bin/bash
# Developer tool (GPU box): per-kernel times of the C4 bench command for prebuilt library variants (tools/build_variants.sh).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for T in "$@"; do
  L=$R/libxaac_amd/libxaac_amd_$T.so
  [ "$T" = base ] && L=$R/libxaac_amd/libxaac_amd.so
  XAAC_AMD_LIBRARY=$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/v_$T -o r -- python $R/bench.py --workload ${XAAC_WORKLOAD:-c4} --hip-streams ${XAAC_HIP_STREAMS:-1} --steps 40 --warmup 4 --no-cpu-baseline --no-secondary > /tmp/v_$T.json 2>/dev/null
  echo "== $T: $(python -c "import json; d=json.load(open('/tmp/v_$T.json')); print(d['ms_per_step'], d['bit_exact_vs_oracle'], d['refused_frac'])")"
  python $R/tools/rocprof_summary.py stats $(find /tmp/v_$T -name "*.db") | sed -n 2,6p | cut -c1-125
done
