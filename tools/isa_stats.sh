#!/bin/bash
# tools/isa_stats.sh [extra hipcc flags] -- compile the IMDCT kernel's hot path alone
# (-DXAAC_HOT_ONLY) and print its register use and instruction histogram.
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
cp libxaac_amd/csrc/*.h libxaac_amd/csrc/*.inc libxaac_amd/csrc/imdct_kernel.hip $T/
mkdir -p $T/../include 2>/dev/null || true
sed -i 's#"../../include/xaac_amd.h"#"'$PWD'/include/xaac_amd.h"#' $T/imdct_kernel.h
(cd $T && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DXAAC_HOT_ONLY "$@" -c imdct_kernel.hip -o k.o -save-temps -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error| VGPRs:|SGPRs:|Scratch|Occupancy" | sed 's/.*remark: //')
S=$T/imdct_kernel-hip-amdgcn-amd-amdhsa-gfx950.s
echo "total instructions: $(grep -cE '^\s+(v_|ds_|global_|s_|buffer_|scratch_|flat_)' $S)"
grep -E '^\s+(v_|ds_|global_|s_waitcnt|s_cbranch|buffer_|scratch_|flat_)' $S | awk '{print $1}' | sort | uniq -c | sort -rn | head -${TOPN:-45}
cp $S /tmp/imdct_hot.s
rm -rf $T
