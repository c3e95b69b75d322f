#!/bin/bash
# Developer tool (GPU box): the reference-made HQ / PS parity tests, then per-kernel times of the C4 bench command for the
# in-tree library and any prebuilt variants (tools/build_variants.sh):   bash tools/ab.sh [variant tags...]
R=$GRAFT_REPO_ROOT
cd $R && timeout 600 python -m pytest tests/test_sbr_hq_gpu.py tests/test_sbr_chains.py -m gpu -x -q 2>&1 | tail -2
bash $R/tools/time_variants.sh base "$@" 2>&1 | grep -E "==|core_kernel|ps_kernel|synthesis|analysis|imdct"
