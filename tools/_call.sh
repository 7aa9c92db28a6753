O=gpurun_out/r03p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 python -m pytest $GRAFT_REPO_ROOT/tests -q -m gpu -x -k "mlp or c1" 2>&1 | tail -3
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --workload c1 --steps 4000 --warmup 200 --no-cpu-baseline > $O/c1_f1.json 2>$O/c1.err; python -c "
import json; d=json.load(open('$O/c1_f1.json')); print('c1', d['value'], d['ms_per_step'], d['roofline'].get('avg_step_device_ms'))"
timeout 300 python tools/soak_c1.py 40000 10000 2>&1 | tail -2
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pc1 -o c1 -- python $GRAFT_REPO_ROOT/bench.py --workload c1 --steps 2000 --warmup 100 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('/tmp/pc1/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:3]: print(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3)
PY
