import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, numpy as np
from neunet_hip import _lib
from neunet_hip._lib import call_hip_function as call
st = _lib.get_current_stream_ptr()
R, D = 8192, 4096; n = R*D
x, dy, y, dx = [torch.randn(R, D, device="cuda") for _ in range(4)]
p, m, v = torch.randn(R, D, device="cuda"), torch.zeros(R, D, device="cuda"), torch.zeros(R, D, device="cuda")
w, std = torch.ones(D, device="cuda"), torch.empty(R, device="cuda")
ops = {
 "swish": lambda: call("nnhipSwishForward", y, x, 1.0, n, st),
 "rms": lambda: call("nnhipRMSNormForward", x, w, None, y, std, None, R, D, 1e-6, st),
 "softmax": lambda: call("nnhipSoftmaxForward", y, x, R, D, 1, st),
 "adamw": lambda: call("nnhipFusedAdamWStep", p, dy, m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 3, n, 0, 1.0, st),
 "relu": lambda: call("nnhipReLUForward", y, x, n, st),
}
def timeit(prev, cur, iters=20):
    ts = []
    for _ in range(iters):
        ops[prev]()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops[cur](); b.record(); ts.append((a, b))
    torch.cuda.synchronize()
    return np.median([a.elapsed_time(b) for a, b in ts]) * 1e3
for cur in ["swish", "rms", "softmax", "relu"]:
    print(cur, " after adamw:", round(timeit("adamw", cur), 1), "us   after softmax:", round(timeit("softmax", cur), 1), "us   after itself:", round(timeit(cur, cur), 1), "us", flush=True)
