#!/bin/bash
# Developer tool (GPU box): ms per step of bench.py's default run (two HIP streams) for library variants, three runs each.
R=$GRAFT_REPO_ROOT
for T in "$@"; do
  L=$R/libxaac_amd/libxaac_amd_$T.so
  [ "$T" = base ] && L=$R/libxaac_amd/libxaac_amd.so
  for i in 1 2 3; do
    XAAC_AMD_LIBRARY=$L python $R/bench.py --workload ${XAAC_WORKLOAD:-c4} --steps 80 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$T', d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact_vs_oracle'], d['refused_frac'])"
  done
done
