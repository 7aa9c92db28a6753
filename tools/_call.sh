cd /tmp; export TMPDIR=/tmp
for v in -1 0; do rm -rf /tmp/pc4
NNHIP_ATTN_PAIR=$v timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pc4 -o c4 -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 8 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('/tmp/pc4/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'attn_fwd' in r['Name']: print('pair=$v', r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['MinNs'])/1e3,1))
PY
done
