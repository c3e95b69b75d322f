#!/bin/bash
# where the IMDCT kernel's cycles go: busy / active / wait counters, <= 4 per pass, each pass under its own timeout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_IFETCH"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d /tmp/pd_$i -o r -- python $R/tools/pmc_probe.py > /dev/null 2>&1 || echo "pass $i failed: $set"
done
python $R/tools/rocprof_summary.py pmc $(find /tmp/pd_* -name "*.db") | grep -i "imdct\|counter" 
