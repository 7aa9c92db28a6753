cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_all.log 2>&1; tail -4 $O/tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('instep',{}).get('frac'), d['gemm_mode'])"
