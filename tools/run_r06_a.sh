cd $GRAFT_REPO_ROOT
bash tools/clock_probe.sh r06_a
export XAAC_HIP_STREAMS=1
bash tools/time_variants.sh base > gpurun_out/r06_a_wg2.txt 2>&1
XAAC_CORE_WG_PER_CU=1 bash tools/time_variants.sh base > gpurun_out/r06_a_wg1.txt 2>&1
cat gpurun_out/r06_a_wg2.txt gpurun_out/r06_a_wg1.txt
