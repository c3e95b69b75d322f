#!/bin/bash
# Round-6 evidence for profiles/ (GPU box): for every configuration bench.py prints a number for, the rocprofv3 kernel
# stats of the command that made it and the SQ counter passes; for C4 and C3 the HBM traffic passes as well (FETCH_SIZE and
# WRITE_SIZE in runs of their own, as the micro-architecture guide prescribes; never together with a trace).
#   bash tools/profile_r06.sh <tag> [driver] [bench] [c4] [c3] [esbr] [f4]      (no section named: all of them)
# writes gpurun_out/<tag>_*; the real VGPR / LDS / scratch figures of the kernels are tools/codeobj_notes.py's (CPU).
TAG=${1:-r06}
shift
SECTIONS=${*:-driver bench c4 c3 esbr f4}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU"
has() { case " $SECTIONS " in *" $1 "*) return 0;; esac; return 1; }
summary() { python $R/tools/rocprof_summary.py "$@"; }

chain() { # $1 = c4 | c3: kernel stats of the bench command, HBM and SQ counter passes over the probe
  W=$1
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_${TAG}_$W -o r -- python $R/bench.py --workload $W --hip-streams 1 --steps 60 --warmup 6 --no-cpu-baseline --no-secondary > $O/${TAG}_${W}_bench_under_rocprof.json 2> /dev/null
  summary stats $(find /tmp/ks_${TAG}_$W -name "*.db") > $O/${TAG}_${W}_kernel_stats.txt
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "$SQ1"; do
    i=$((i+1))
    WORKLOAD=$W timeout 600 rocprofv3 --pmc $C -d /tmp/pmc_${TAG}_${W}_$i -o r -- python $R/tools/pmc_probe_sbr.py > /dev/null 2>&1 || echo "pmc pass $i of $W failed"
  done
  N=$([ $W = c4 ] && echo sbr || echo c3)
  summary pmc $(find /tmp/pmc_${TAG}_${W}_1 /tmp/pmc_${TAG}_${W}_2 -name "*.db") > $O/${TAG}_${N}_pmc_hbm.txt
  summary pmc $(find /tmp/pmc_${TAG}_${W}_3 -name "*.db") > $O/${TAG}_${N}_pmc_sq.txt
  head -8 $O/${TAG}_${W}_kernel_stats.txt | cut -c1-130
}

# the library these passes run on (tools/pmc_to_json.py stamps profiles/pmc_latest.json with it; bench.py compares)
python -c "import sys; sys.path.insert(0, '$R'); import libxaac_amd; v = libxaac_amd.load_library().xaac_version().decode(); print(v.split(' build ')[1])" > $O/${TAG}_build_id.txt
if has driver; then # the driver's own command line, next to the long one (same box, alternating)
  for rep in 1 2; do
    python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null > $O/${TAG}_bench_c4_driver_args_$rep.json
    python $R/bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null > $O/${TAG}_bench_c4_long_$rep.json
  done
  python -c "
import json
for n in ('driver_args_1', 'long_1', 'driver_args_2', 'long_2'):
    d = json.load(open('$O/${TAG}_bench_c4_%s.json' % n)); print(n, d['steps'], d['warmup'], d['ms_per_step'], d['roofline']['frac'])"
fi
if has bench; then
  python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_c4.json 2> $O/${TAG}_bench_c4.err
  python -c "import json; d=json.load(open('$O/${TAG}_bench_c4.json')); s=d['secondary']; print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['bit_exact_vs_oracle'], d['refused_frac'], {k: (v.get('ms_per_step'), v.get('bit_exact_vs_oracle')) for k, v in s.items()}, s.get('c4_batch_sweep'), {k: v.get('roofline_frac') for k, v in s['f4_transforms'].items() if isinstance(v, dict)}, d['cpu_baseline']['value'])"
fi
has c4 && chain c4
has c3 && chain c3
if has esbr; then # Path A: bench.py's secondary_esbr alone (with and without the harmonic transposer)
  STEPS=20 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_${TAG}_esbr -o r -- python $R/tools/prof_esbr.py > $O/${TAG}_esbr_bench_under_rocprof.json 2> /dev/null
  summary stats $(find /tmp/ks_${TAG}_esbr -name "*.db") > $O/${TAG}_esbr_kernel_stats.txt
  STEPS=6 timeout 300 rocprofv3 --pmc $SQ1 -d /tmp/pmc_${TAG}_esbr -o r -- python $R/tools/prof_esbr.py > /dev/null 2>&1 || echo "pmc pass of esbr failed"
  summary pmc $(find /tmp/pmc_${TAG}_esbr -name "*.db") > $O/${TAG}_esbr_pmc_sq.txt
  head -10 $O/${TAG}_esbr_kernel_stats.txt | cut -c1-130
fi
if has f4; then # the f4 transforms: bench.py's secondary_f4 alone
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_${TAG}_f4 -o r -- python $R/tools/pmc_probe_f4.py > $O/${TAG}_f4_bench_under_rocprof.txt 2> /dev/null
  summary stats $(find /tmp/ks_${TAG}_f4 -name "*.db") > $O/${TAG}_f4_kernel_stats.txt
  timeout 300 rocprofv3 --pmc $SQ1 -d /tmp/pmc_${TAG}_f4 -o r -- python $R/tools/pmc_probe_f4.py > /dev/null 2>&1 || echo "pmc pass of f4 failed"
  summary pmc $(find /tmp/pmc_${TAG}_f4 -name "*.db") > $O/${TAG}_f4_pmc_sq.txt
  head -12 $O/${TAG}_f4_kernel_stats.txt | cut -c1-130
fi
