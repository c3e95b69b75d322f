cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python $R/bench.py --workload c2l > $R/gpurun_out/bench_c2l.json 2> $R/gpurun_out/bench_c2l.err
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_c2l -o r -- python $R/bench.py --workload c2l --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocprof_summary.py stats $(find /tmp/ks_c2l -name "*.db") > $R/gpurun_out/c2l_kernel_stats.txt
bash $R/tools/pmc_limiter.sh > $R/gpurun_out/limiter_pmc.txt 2>&1
python $R/__graft_entry__.py smoke 2>&1 | tail -4
tail -c 600 $R/gpurun_out/bench_c2l.json
