cd /tmp; export TMPDIR=/tmp
timeout 2400 python -m pytest $GRAFT_REPO_ROOT/tests -q -m gpu -x 2>&1 | tail -5
cd $GRAFT_REPO_ROOT
python tools/kbench.py --only wgroup --iters 40 2>&1 | grep "linear"
python bench.py --workload c4 --no-cpu-baseline --force-dp > gpurun_out/c4_dp.json 2>gpurun_out/c4_dp.err; python -c "
import json; d=json.load(open('gpurun_out/c4_dp.json')); print('forced dp', d['value'], d['ms_per_step'], d['config'].get('dp_mode'))"
python bench.py --no-cpu-baseline > gpurun_out/head.json 2>gpurun_out/head.err; python -c "
import json; d=json.load(open('gpurun_out/head.json')); print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])"
