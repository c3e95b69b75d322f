#!/bin/bash
# where the Path A kernels' cycles go: SQ counters, <= 4 per pass, each pass under its own timeout (tools/prof_esbr.py, 8192 streams)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES"; do
  i=$((i+1))
  STEPS=6 timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pes_$i -o r -- python $R/tools/prof_esbr.py > /dev/null 2>&1 || echo "pass $i failed: $set"
done
python $R/tools/rocprof_summary.py pmc $(find /tmp/pes_* -name "*.db") | grep -i "esbr\|hbe\|counter"
