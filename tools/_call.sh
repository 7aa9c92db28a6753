cd $GRAFT_REPO_ROOT
S=$(date +%s); python bench.py > gpurun_out/default.json 2> gpurun_out/default.err; E=$(date +%s); echo "elapsed $((E-S)) s"; wc -l gpurun_out/default.json; python -c "
import json; d=json.load(open('gpurun_out/default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])"
