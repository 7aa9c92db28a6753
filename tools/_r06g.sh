cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
for S in 0 1 0 1; do NNHIP_CE_REVERSE=$S timeout 600 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_r$S.json 2> $O/bench_c4_r$S.err; python -c "
import json; d=json.load(open('$O/bench_c4_r$S.json')); print('c4 ce reverse $S', d['ms_per_step'])" || tail -5 $O/bench_c4_r$S.err; done
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_all.log 2>&1; tail -5 $O/tests_all.log
timeout 300 tools/probes/stream_nm_probe 2>&1 | tee $O/stream_nm_probe.txt
