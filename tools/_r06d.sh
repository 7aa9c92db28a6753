cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
timeout 300 python tools/asum_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/asum_ab.txt
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "counted_waits or balanced_t256 or waiting_for_the_optimizer" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('c4', d['ms_per_step'])"
timeout 600 python bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline --force-dp --dp-op avg > $O/bench_c4_dp.json 2> $O/bench_c4_dp.err; python -c "
import json; d=json.load(open('$O/bench_c4_dp.json')); print('c4 dp', d['ms_per_step'], d.get('allreduce_ms'), d.get('rccl'))"
