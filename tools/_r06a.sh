cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "device_error or waiting_for_the_optimizer or error_status" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('c4', d['ms_per_step'])"
timeout 300 python tools/find_torch_nodes.py > $O/torch_nodes.txt 2>&1; tail -30 $O/torch_nodes.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/$O/trace_c4 -o c4 -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/trace_c4.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r06a/trace_c4/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# keep the last step only: find the last adamw_multi and the one before
idx = [i for i, r in enumerate(rows) if 'adamw_multi' in r['Kernel_Name']]
lo, hi = idx[-2] + 1, idx[-1] + 1
out = open('gpurun_out/r06a/last_step.tsv', 'w')
prev_end = None
for r in rows[lo:hi]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = r['Kernel_Name'].replace('void nnhip::', '')[:70]
    out.write(f"{(e - s) / 1e3:9.1f}\t{gap:7.1f}\t{r['Grid_Size']}\t{r['Workgroup_Size']}\t{name}\n")
    prev_end = e
out.close()
import os
os.remove(f)
PY
find $O -name "*.db" -delete; du -sh $O
