O=gpurun_out/r03o; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
python bench.py --workload nb --no-cpu-baseline > $O/nb.json 2>$O/nb.err; tail -3 $O/nb.err; python -c "
import json; d=json.load(open('$O/nb.json')); print(d['it_per_s'], d['it_per_s_eager'], d['vs_reference_notebook'], d['config']['launch'], d['ms_per_step'])"
