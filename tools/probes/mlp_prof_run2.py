import ctypes, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from soak_c1 import build
from neunet_hip import _lib
rng = np.random.default_rng(3)
X = rng.uniform(-1, 1, (4, 32, 784)).astype(np.float32); Y = rng.integers(0, 10, (4, 32)).astype(np.int32)
step, ps, opt, x, y = build(True, X, Y)
for _ in range(50): step()
torch.cuda.synchronize()
lib = _lib.load_library()
f = lib.nnhipDebugMlpProfRead; f.argtypes = [ctypes.c_void_p, ctypes.c_int]; f.restype = ctypes.c_int
buf = np.zeros(512 * 8, dtype=np.int64); f(buf.ctypes.data, 512 * 8)
t = buf.reshape(512, 8)
r = (t[8:400, 1:8] - t[8:400, 0:1]).astype(np.float64)
names = ["loads+state", "W2 slice in LDS", "dZ built", "MFMA done", "reduce barrier", "adam+stores issued", "stores landed"]
for k, nm in enumerate(names):
    print(f"{nm:20s} median {np.median(r[:, k]):7.0f}  p90 {np.percentile(r[:, k], 90):7.0f}  max {r[:, k].max():7.0f} ticks")
