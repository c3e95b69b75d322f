#!/bin/bash
# SQ counters of the C4 chain's kernels (one rocprofv3 --pmc pass over tools/pmc_probe_sbr.py).  Usage on the GPU box:
#   bash tools/pmc_sq.sh <tag>      -> gpurun_out/<tag>_sbr_pmc_sq.txt
TAG=${1:-sq}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU -d /tmp/pmc_${TAG}_3 -o r -- python $R/tools/pmc_probe_sbr.py > /dev/null 2>&1
python $R/tools/rocprof_summary.py pmc $(find /tmp/pmc_${TAG}_3 -name "*.db") > $R/gpurun_out/${TAG}_sbr_pmc_sq.txt
grep -E "core|ps_kernel|synthesis_pair" $R/gpurun_out/${TAG}_sbr_pmc_sq.txt | cut -c1-140
