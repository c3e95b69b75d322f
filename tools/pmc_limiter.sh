#!/bin/bash
# rocprofv3 counter passes over the limiter kernel (tools/time_limiter.py), one signal kind at a time.
# <= 4 SQ counters a pass, one TCC counter a pass; every pass under its own timeout.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kind in quiet loud; do
  XL_KIND=$kind timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/pl_$kind -o r -- python $R/tools/time_limiter.py base > /tmp/pl_$kind.log 2>&1 || echo "sq pass failed ($kind)"
  for c in FETCH_SIZE WRITE_SIZE; do
    XL_KIND=$kind timeout 150 rocprofv3 --kernel-trace --pmc $c -d /tmp/pm_${kind}_$c -o r -- python $R/tools/time_limiter.py base > /tmp/pm_$kind.log 2>&1 || echo "$c pass failed ($kind)"
  done
  echo "== $kind"
  python $R/tools/rocprof_summary.py pmc $(find /tmp/pl_$kind /tmp/pm_${kind}_* -name "*.db") | grep -i "limiter\|counter"
done
