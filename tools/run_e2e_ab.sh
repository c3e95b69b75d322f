cd $GRAFT_REPO_ROOT
for S in 2; do
python bench.py --hip-streams $S --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['secondary']['end_to_end']
print('streams $S', d['ms_per_step'], '| e2e', e['value'], e['gpu_and_copies_s'], '| esbr', e['esbr']['value'], e['esbr']['parse_s'], e['esbr']['gpu_and_copies_s'], e['esbr']['wall_s'], '| pathA', d['secondary']['c4_esbr']['ms_per_step'])"
done
