cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "wave_private" > $O/tests.log 2>&1; tail -2 $O/tests.log
for M in stream pw stream pw stream pw; do echo -n "$M: "; NNHIP_ATTN_SB_FWD=$M timeout 300 python tools/attn_sb_time.py 2>&1 | grep -v amdgpu | tail -1; done
for M in stream pw stream pw; do NNHIP_ATTN_SB_FWD=$M timeout 600 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_$M.json 2> $O/bench_c4_$M.err; python -c "
import json; d=json.load(open('$O/bench_c4_$M.json')); print('c4 attn fwd $M', d['ms_per_step'])"; done
