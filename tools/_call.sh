O=gpurun_out/r03p; mkdir -p $O
for f in 1 0; do
NNHIP_C1_FUSE_OPT=$f python bench.py --workload c1 --steps 4000 --warmup 200 --no-cpu-baseline > $O/c1_f$f.json 2>$O/c1.err; python -c "
import json; d=json.load(open('$O/c1_f$f.json')); print('fuse=$f', d['value'], d['ms_per_step'], d['roofline'].get('avg_step_device_ms'))"
done
NNHIP_MLP_CHAIN=0 NNHIP_C1_FUSE_OPT=0 python bench.py --workload c1 --steps 4000 --warmup 200 --no-cpu-baseline > $O/c1_nc.json 2>$O/c1.err; python -c "
import json; d=json.load(open('$O/c1_nc.json')); print('nochain', d['value'], d['ms_per_step'], d['roofline'].get('avg_step_device_ms'))"
cd /tmp; export TMPDIR=/tmp
timeout 600 python -m pytest $GRAFT_REPO_ROOT/tests -q -m gpu -x -k "mlp" 2>&1 | tail -3
rocprofv3 --kernel-trace --stats -f csv -d /tmp/pc1 -o c1 -- python $GRAFT_REPO_ROOT/bench.py --workload c1 --steps 2000 --warmup 100 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('/tmp/pc1/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3)
PY
