cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/pc4 -o c4 -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('/tmp/pc4/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]: print(r['Name'][:90], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
