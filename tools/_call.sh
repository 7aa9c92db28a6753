cd /tmp; export TMPDIR=/tmp
for i in 1 2; do timeout 2400 python -m pytest $GRAFT_REPO_ROOT/tests -q -m gpu -x 2>&1 | tail -2; done
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline > gpurun_out/head.json 2>gpurun_out/head.err; python -c "
import json; d=json.load(open('gpurun_out/head.json')); print('headline', d['value'], d['ms_per_step'], d['roofline']['frac']); print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d.get('also',{}).items() if isinstance(v,dict) and 'value' in v})"
