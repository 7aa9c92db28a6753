cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "conv" -x 2>&1 | tail -4
for F in 0 1; do
NNHIP_CONV_BWD_FORK=$F python bench.py --workload c5 --steps 4800 --warmup 320 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('fork=$F c5', d['value'], d['ms_per_step'], d['roofline']['avg_step_device_ms'], d['launches_per_step'])"
done
