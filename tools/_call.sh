O=gpurun_out/r03p; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for v in "" head; do
L=$GRAFT_REPO_ROOT/numpy-nn-model_amd/neunet_hip/lib/libneunet_hip${v:+.$v}.so
NEUNET_HIP_LIB=$L timeout 300 python bench.py --workload c1 --steps 6000 --warmup 300 --no-cpu-baseline > $O/c1_$v.json 2>$O/c1.err; python -c "
import json; d=json.load(open('$O/c1_$v.json')); print('c1 lib=$v', d['value'], d['ms_per_step'])"
done; done
