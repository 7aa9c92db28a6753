O=gpurun_out/r03p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 1200 python -m pytest $GRAFT_REPO_ROOT/tests -q -m gpu -x -k "conv" 2>&1 | tail -2
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pc5 -o c5 -- python $GRAFT_REPO_ROOT/bench.py --workload c5 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('/tmp/pc5/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    if 'conv' in r['Name']: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
