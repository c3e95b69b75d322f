#!/bin/bash
# C2 evidence for profiles/: bench JSON, rocprofv3 kernel stats of the same command, SQ and HBM counter passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python $R/bench.py > $R/gpurun_out/bench_c2.json 2> $R/gpurun_out/bench_c2.err
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_c2 -o r -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocprof_summary.py stats $(find /tmp/ks_c2 -name "*.db") > $R/gpurun_out/c2_kernel_stats.txt
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/sq_c2 -o r -- python $R/tools/pmc_probe.py > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/hbm_c2_$c -o r -- python $R/tools/pmc_probe.py > /dev/null 2>&1
done
python $R/tools/rocprof_summary.py pmc $(find /tmp/sq_c2 /tmp/hbm_c2_* -name "*.db") > $R/gpurun_out/c2_pmc.txt
head -4 $R/gpurun_out/c2_kernel_stats.txt; grep -i "imdct\|copy" $R/gpurun_out/c2_pmc.txt; cut -c1-330 $R/gpurun_out/bench_c2.json
