cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "balanced or counted or attention or gpt" > $O/tests.log 2>&1; tail -4 $O/tests.log
for M in stream lds stream lds; do NNHIP_ATTN_SB_FWD=$M timeout 300 python tools/attn_sb_time.py 2>&1 | grep -v amdgpu | tail -3; done
for M in stream lds stream lds; do NNHIP_ATTN_SB_FWD=$M timeout 600 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_$M.json 2> $O/bench_c4_$M.err; python -c "
import json; d=json.load(open('$O/bench_c4_$M.json')); print('c4 attn fwd $M', d['ms_per_step'])" || tail -5 $O/bench_c4_$M.err; done
