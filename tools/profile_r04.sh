#!/bin/bash
# Round-4 evidence for profiles/: the driver-style bench line (C4 default), rocprofv3 kernel stats of the same command,
# and PMC passes (HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes; SQ: issue / lane-use counters) over
# tools/pmc_probe_sbr.py.  Usage on the GPU box: bash tools/profile_r03.sh <tag>   (writes gpurun_out/<tag>_*)
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python $R/bench.py > $R/gpurun_out/${TAG}_bench_c4.json 2> $R/gpurun_out/${TAG}_bench_c4.err
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_$TAG -o r -- python $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python $R/tools/rocprof_summary.py stats $(find /tmp/ks_$TAG -name "*.db") > $R/gpurun_out/${TAG}_c4_kernel_stats.txt
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C -d /tmp/pmc_${TAG}_$i -o r -- python $R/tools/pmc_probe_sbr.py > /dev/null 2>&1
done
python $R/tools/rocprof_summary.py pmc $(find /tmp/pmc_${TAG}_1 /tmp/pmc_${TAG}_2 -name "*.db") > $R/gpurun_out/${TAG}_sbr_pmc_hbm.txt
python $R/tools/rocprof_summary.py pmc $(find /tmp/pmc_${TAG}_3 -name "*.db") > $R/gpurun_out/${TAG}_sbr_pmc_sq.txt
head -9 $R/gpurun_out/${TAG}_c4_kernel_stats.txt
python -c "import json; d=json.load(open('$R/gpurun_out/${TAG}_bench_c4.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['bit_exact_vs_oracle'], d['refused_frac'], {k: (v.get('value'), v.get('bit_exact_vs_oracle')) for k, v in d['secondary'].items()}, d['cpu_baseline']['value'])"
