#!/bin/bash
# C3 / C4 evidence for profiles/: bench JSONs and rocprofv3 kernel stats of the same commands.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for w in c3 c4; do
  python $R/bench.py --workload $w > $R/gpurun_out/bench_$w.json 2> $R/gpurun_out/bench_$w.err
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$w -o r -- python $R/bench.py --workload $w --steps 60 --warmup 6 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/rocprof_summary.py stats $(find /tmp/ks_$w -name "*.db") > $R/gpurun_out/${w}_kernel_stats.txt
  head -8 $R/gpurun_out/${w}_kernel_stats.txt
  python -c "import json; d=json.load(open('$R/gpurun_out/bench_$w.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
done
