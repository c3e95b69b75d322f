cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
echo "== base"; python tools/time_c4_alternate.py 2 2>&1 | tail -2
echo "== ps64"; XAAC_AMD_LIBRARY=$R/libxaac_amd/libxaac_amd_ps64.so python tools/time_c4_alternate.py 2 2>&1 | tail -2
echo "== ps16"; XAAC_AMD_LIBRARY=$R/libxaac_amd/libxaac_amd_ps16.so python tools/time_c4_alternate.py 2 2>&1 | tail -2
echo "== core1 ps1"; XAAC_CORE_WG_PER_CU=1 XAAC_PS_WG_PER_CU=1 python tools/time_c4_alternate.py 2 2>&1 | tail -2
echo "== core1 ps2"; XAAC_CORE_WG_PER_CU=1 python tools/time_c4_alternate.py 2 2>&1 | tail -2
echo "== ps64 core1 ps1"; XAAC_CORE_WG_PER_CU=1 XAAC_PS_WG_PER_CU=1 XAAC_AMD_LIBRARY=$R/libxaac_amd/libxaac_amd_ps64.so python tools/time_c4_alternate.py 2 2>&1 | tail -2
echo "== bench 1 vs 2 streams"
python bench.py --hip-streams 1 --steps 60 --warmup 6 --no-cpu-baseline --no-secondary | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact_vs_oracle'], d['refused_frac'])"
python bench.py --hip-streams 2 --steps 60 --warmup 6 --no-cpu-baseline --no-secondary | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact_vs_oracle'], d['refused_frac'])"
python bench.py --workload c3 --hip-streams 1 --steps 40 --warmup 6 --no-cpu-baseline --no-secondary | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact_vs_oracle'], d['refused_frac'])"
python bench.py --workload c3 --hip-streams 2 --steps 40 --warmup 6 --no-cpu-baseline --no-secondary | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact_vs_oracle'], d['refused_frac'])"
python bench.py --workload c2 --hip-streams 1 --steps 100 --warmup 6 --no-cpu-baseline --no-secondary | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact_vs_oracle'], d['refused_frac'])"
python bench.py --workload c2 --hip-streams 2 --steps 100 --warmup 6 --no-cpu-baseline --no-secondary | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact_vs_oracle'], d['refused_frac'])"
