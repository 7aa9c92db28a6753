cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "persistent or device_error or waiting_for_the_optimizer or linear_swish" > $O/tests.log 2>&1; tail -3 $O/tests.log
for S in 0 1 0 1; do NNHIP_PST_STAGGER=$S timeout 300 python tools/pst_ab.py > $O/pst_ab_$S.json 2>$O/pst_ab_$S.err; python - <<PY
import json; d=json.load(open('$O/pst_ab_$S.json')); print('stagger', d['stagger'], {k: v['us'] for k, v in d.items() if k != 'stagger'})
PY
done
python - <<PY
import json
a, b = json.load(open('$O/pst_ab_0.json')), json.load(open('$O/pst_ab_1.json'))
print('bit-identical:', all(a[k]['sha'] == b[k]['sha'] for k in a if k != 'stagger'))
PY
for S in 0 1 0 1; do NNHIP_PST_STAGGER=$S timeout 600 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_s$S.json 2> $O/bench_c4_s$S.err; python -c "
import json; d=json.load(open('$O/bench_c4_s$S.json')); print('c4 stagger $S', d['ms_per_step'])"; done
