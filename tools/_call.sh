O=gpurun_out/r03p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 1200 python -m pytest $GRAFT_REPO_ROOT/tests -q -m gpu -x -k "linear or conv_classifier or small or pair or mlp" 2>&1 | tail -2
cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 300 python bench.py --workload c5 --no-cpu-baseline > $O/c5.json 2>$O/c5.err; python -c "
import json; d=json.load(open('$O/c5.json')); print('c5', d['value'], d['ms_per_step'])"; done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pc5 -o c5 -- python $GRAFT_REPO_ROOT/bench.py --workload c5 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('/tmp/pc5/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    if 'small' in r['Name']: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
