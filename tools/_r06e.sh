cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "swish or persistent or ffn or gpt or golden" > $O/tests.log 2>&1; tail -3 $O/tests.log
for S in 1 2; do PST_AB_SWISH_MODE=$S timeout 300 python tools/pst_ab.py > $O/pst_ab_m$S.json 2>$O/pst_ab_m$S.err; python - <<PY
import json; d=json.load(open('$O/pst_ab_m$S.json')); print('swish mode $S', {k: v['us'] for k, v in d.items() if k != 'stagger'})
PY
done
for S in 0 1 0 1; do NNHIP_SWISH_SAVE_DERIVATIVE=$S timeout 600 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_d$S.json 2> $O/bench_c4_d$S.err; python -c "
import json; d=json.load(open('$O/bench_c4_d$S.json')); print('c4 save_derivative $S', d['ms_per_step'])"; done
