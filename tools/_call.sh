cd /tmp; export TMPDIR=/tmp
timeout 2400 python -m pytest $GRAFT_REPO_ROOT/tests -q -m gpu -x -p no:cacheprovider 2>&1 | tail -1
cd $GRAFT_REPO_ROOT
python bench.py --workload c4 --no-cpu-baseline > gpurun_out/c4_x.json 2>gpurun_out/c4.err; python -c "
import json; d=json.load(open('gpurun_out/c4_x.json')); print('c4', d['value'], d['ms_per_step'])"
