#!/bin/bash
# Developer tool (GPU box): a few more SQ / TCP counters of the C4 chain's kernels, several passes (tools/pmc_probe_sbr.py).
#   bash tools/pmc_detail.sh <tag>      -> gpurun_out/<tag>_pmc_detail.txt
TAG=${1:-det}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE" \
         "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C -d /tmp/pmcd_${TAG}_$i -o r -- python $R/tools/pmc_probe_sbr.py > /tmp/pmcd_$i.log 2>&1 || tail -3 /tmp/pmcd_$i.log
done
python $R/tools/rocprof_summary.py pmc $(find /tmp/pmcd_${TAG}_* -name "*.db") > $R/gpurun_out/${TAG}_pmc_detail.txt
grep -E "core_kernel<1|ps_kernel|synthesis_pair" $R/gpurun_out/${TAG}_pmc_detail.txt | cut -c1-140
