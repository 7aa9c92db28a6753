cd /tmp; export TMPDIR=/tmp
timeout 2400 python -m pytest $GRAFT_REPO_ROOT/tests -q -m gpu -x -p no:cacheprovider 2>&1 | tail -1
timeout 600 python -c "import sys; sys.path.insert(0,'$GRAFT_REPO_ROOT'); import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline > gpurun_out/head.json 2>gpurun_out/head.err; python -c "
import json; d=json.load(open('gpurun_out/head.json')); print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])"
