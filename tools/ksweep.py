import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from neunet_hip import _lib
from neunet_hip._lib import call_hip_function as call
from kbench import bench
st = _lib.get_current_stream_ptr()
for (M, N) in [(4096, 4096)]:
    for K in [512, 4096]:
        X = torch.rand(M, K, device="cuda") - 0.5; W = torch.rand(N, K, device="cuda") - 0.5; O = torch.empty(M, N, device="cuda")
        med, mn = bench(lambda: call("nnhipLinearModuleForward", X, W, None, O, M, K, N, st), 20)
        fl = 2.0 * M * N * K
        print(f"M={M} N={N} K={K:5d}: {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF   per-iter(32) {med*1e3/(K/32):6.2f} us", flush=True)
