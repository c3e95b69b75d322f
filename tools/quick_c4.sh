#!/bin/bash
# Quick C4 iteration loop on the GPU box: the parity tests that meet the C4 chain, then kernel stats of the bench command.
# Usage: bash tools/quick_c4.sh <tag> [pytest-k-expression]
TAG=${1:-q}
K=${2:-"sbr or qmf or dropin or batch_host"}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R && timeout 900 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_$TAG -o r -- python $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_bench_c4.json 2> $R/gpurun_out/${TAG}_bench.err
python $R/tools/rocprof_summary.py stats $(find /tmp/ks_$TAG -name "*.db") > $R/gpurun_out/${TAG}_c4_kernel_stats.txt
head -12 $R/gpurun_out/${TAG}_c4_kernel_stats.txt
python -c "import json; d=json.load(open('$R/gpurun_out/${TAG}_bench_c4.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['bit_exact_vs_oracle'], d['refused_frac'])"
