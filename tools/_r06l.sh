cd $GRAFT_REPO_ROOT
O=gpurun_out/r06l; mkdir -p $O
run() { # name, env...
  name=$1; shift
  for W in c4 c1 c5; do
    S=20; WU=5; [ $W = c1 ] && S=512 && WU=32; [ $W = c5 ] && S=480 && WU=32
    env "$@" timeout 600 python bench.py --workload $W --steps $S --warmup $WU --no-cpu-baseline > $O/${name}_$W.json 2> $O/${name}_$W.err
    python -c "
import json; d=json.load(open('$O/${name}_$W.json')); print('$name', '$W', d['ms_per_step'])" || tail -3 $O/${name}_$W.err
  done
}
run base A=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run base2 A=1
