cd $GRAFT_REPO_ROOT
for cfg in "8 256" "16 256" "32 512" "64 1024"; do set -- $cfg
NNHIP_SMALL_T128=$1 NNHIP_SMALL_T32=$2 python bench.py --workload nb --no-cpu-baseline > gpurun_out/nb_x.json 2>gpurun_out/nb.err; python -c "
import json; d=json.load(open('gpurun_out/nb_x.json')); print('t128=$1 t32=$2', d['value'], d['ms_per_step'], d.get('it_per_s'))"
done
