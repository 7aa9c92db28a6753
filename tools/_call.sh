cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
( time python bench.py > gpurun_out/r04e/bench_headline.json 2> gpurun_out/r04e/bench_headline.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('gpurun_out/r04e/bench_headline.json')); a=d['also']
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print('c1', a['c1']['samples_per_s'], a['c1']['launch'], a['c1']['with_input_copy']['samples_per_s'], a['c1']['separate_optimizer_launch']['samples_per_s'])
print('c5', a['c5']['ms_per_step'])"
