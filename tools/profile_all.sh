#!/bin/bash
# every workload's evidence in one GPU call: bench JSONs + rocprofv3 kernel stats (C2 also the counter passes)
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_c2.sh > /dev/null 2>&1
bash $R/tools/profile_c2l.sh > /dev/null 2>&1
bash $R/tools/profile_sbr.sh > /dev/null 2>&1
for w in c2 c2l c3 c4; do python -c "import json; d=json.load(open('$R/gpurun_out/bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['cpu_baseline']['value'], d['cpu_baseline'].get('value_1core'))"; done
