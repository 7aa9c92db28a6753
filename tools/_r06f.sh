cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
for S in 0 1 0 1; do NNHIP_WGRAD_STREAM=$S timeout 600 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_w$S.json 2> $O/bench_c4_w$S.err; python -c "
import json; d=json.load(open('$O/bench_c4_w$S.json')); print('c4 wgrad side stream $S', d['ms_per_step'], d['config']['launch'])" || tail -5 $O/bench_c4_w$S.err; done
NNHIP_WGRAD_STREAM=1 timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gpt_c4 or graphed" > $O/tests_side.log 2>&1; tail -3 $O/tests_side.log
NNHIP_WGRAD_STREAM=1 timeout 600 python bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline --graph 0 > $O/bench_c4_w1_eager.json 2> $O/bench_c4_w1_eager.err; python -c "
import json; d=json.load(open('$O/bench_c4_w1_eager.json')); print('c4 eager side', d['ms_per_step'])"
timeout 300 tools/probes/stream_nm_probe 2>&1 | tee $O/stream_nm_probe.txt
