cd $GRAFT_REPO_ROOT
NNHIP_ALLOW_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 2 --workload c4 --steps 3 --warmup 1 > gpurun_out/c4_2rank.json 2> gpurun_out/c4_2rank.err; echo rc=$?; tail -c 1500 gpurun_out/c4_2rank.json | cut -c1-1200; tail -5 gpurun_out/c4_2rank.err | cut -c1-300
timeout 600 python bench.py --workload c4 --force-dp --dp-ingraph 1 --no-cpu-baseline > gpurun_out/c4_ing.json 2>gpurun_out/c4_ing.err; python -c "
import json; d=json.load(open('gpurun_out/c4_ing.json')); print('ingraph', d['value'], d['ms_per_step'], d.get('dp_mode'))"
