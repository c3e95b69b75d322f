#!/bin/bash
# Round-6 probe (GPU box): is the driver's line (--steps 20 --warmup 5) the same clock as the long one (--steps 100 --warmup 10)?
# Alternates the two commands (plus a long warm-up in front of 20 steps) so that box drift shows as drift, not as a difference.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=${1:-r06_a}
cd $R
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s ms_per_step %.4f kernel_ms %.4f frac %.4f' % ('$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
{
for rep in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | line "steps20_warm5"
  python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | line "steps100_warm10"
  python bench.py --gpus 1 --steps 20 --warmup 60 --no-cpu-baseline --no-secondary 2>/dev/null | line "steps20_warm60"
  python bench.py --gpus 1 --steps 400 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | line "steps400_warm10"
done
} | tee $O/${TAG}_clock_probe.txt
