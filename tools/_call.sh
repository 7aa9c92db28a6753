cd $GRAFT_REPO_ROOT
NEUNET_HIP_LIB=$GRAFT_REPO_ROOT/numpy-nn-model_amd/neunet_hip/lib/libneunet_hip.mlpprof.so timeout 300 python tools/probes/mlp_prof_run.py 2>&1 | tail -9
