cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
OLD=$GRAFT_REPO_ROOT/numpy-nn-model_amd/neunet_hip/lib/libneunet_hip.pstold.so
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "persistent or swish or ffn" > $O/tests.log 2>&1; tail -3 $O/tests.log
for L in old new old new; do
  if [ $L = old ]; then export NEUNET_HIP_LIB=$OLD; else unset NEUNET_HIP_LIB; fi
  timeout 300 python tools/pst_ab.py > $O/pst_$L.json 2>$O/pst_$L.err; python - <<PY
import json; d=json.load(open('$O/pst_$L.json')); print('$L', {k: v['us'] for k, v in d.items() if k != 'stagger'})
PY
done
python - <<PY
import json
a, b = json.load(open('$O/pst_old.json')), json.load(open('$O/pst_new.json'))
print('bit-identical old vs new:', all(a[k]['sha'] == b[k]['sha'] for k in a if k != 'stagger'))
PY
for L in old new old new; do
  if [ $L = old ]; then export NEUNET_HIP_LIB=$OLD; else unset NEUNET_HIP_LIB; fi
  timeout 600 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_$L.json 2> $O/bench_c4_$L.err; python -c "
import json; d=json.load(open('$O/bench_c4_$L.json')); print('c4 $L', d['ms_per_step'])"; done
unset NEUNET_HIP_LIB
bash tools/_r06l.sh
