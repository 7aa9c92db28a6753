python tools/ce_parts.py 2>&1 | grep -v amdgpu.ids
NEUNET_HIP_LIB=$PWD/numpy-nn-model_amd/neunet_hip/lib/libneunet_hip.base.so python tools/ce_parts.py 2>&1 | grep -v amdgpu.ids
