"""Developer tool: K sweep of the forward GEMM at fixed M x N -- the intercept of time(K) is the fixed cost per generation
of tiles (launch + prologue + epilogue), the slope the k-loop rate.   python tools/ksweep.py [M N]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neunet_hip import _lib  # noqa: E402
from neunet_hip._lib import call_hip_function as call  # noqa: E402
from kbench import bench  # noqa: E402

st = _lib.get_current_stream_ptr()
shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(16384, 512), (16384, 2048), (4096, 4096)]
for (M, N) in shapes:
    ks, ts = [], []
    for K in [32, 64, 128, 256, 512, 1024, 2048]:
        X = torch.rand(M, K, device="cuda") - 0.5
        W = torch.rand(N, K, device="cuda") - 0.5
        O = torch.empty(M, N, device="cuda")
        med, mn = bench(lambda: call("nnhipLinearModuleForward", X, W, None, O, M, K, N, st), 30)
        fl = 2.0 * M * N * K
        ks.append(K); ts.append(med * 1e3)
        print(f"M={M} N={N} K={K:5d}: {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF   per k-tile(32) {med*1e3/(K/32):6.2f} us", flush=True)
    a, b = np.polyfit(ks[2:], ts[2:], 1)
    gens = -(-(M // 128) * -(-N // 128) // 512)
    print(f"   fit K>=128: {b:6.1f} us + {a*32:6.3f} us per k-tile; {gens} generation(s) of <=512 tiles -> {b/gens:5.1f} us fixed per generation; "
          f"k-loop rate {2.0*M*N*32/(a*32*1e-6)/1e12:6.1f} TFLOP/s", flush=True)
