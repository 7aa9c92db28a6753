#!/bin/bash
# developer tool: the attention timings of every library variant under neunet_hip/lib (build.py --variant NAME -D ...)
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in $R/numpy-nn-model_amd/neunet_hip/lib/libneunet_hip*.so; do
  NEUNET_HIP_LIB=$lib timeout 120 python $R/tools/attn_sb_time.py 2>&1 | grep -v amdgpu.ids
done
