cd $GRAFT_REPO_ROOT
python tools/kbench.py --only attn 2>&1 | grep "T256 H8 dh64"
for n in 1 2 4; do echo skew$n; NEUNET_HIP_LIB=$PWD/numpy-nn-model_amd/neunet_hip/lib/libneunet_hip.skew$n.so python tools/kbench.py --only attn 2>&1 | grep "fwd B64 T256 H8 dh64"; done
