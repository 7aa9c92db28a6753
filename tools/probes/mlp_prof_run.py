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
buf = np.zeros(512 * 8, dtype=np.int64)
print("rc", f(buf.ctypes.data, 512 * 8))
t = buf.reshape(512, 8)
nb = int((t[:, 0] != 0).sum())
d = t[:nb, 1:5] - t[:nb, 0:1]          # ticks since the block's own start (clocks of different XCDs are not comparable)
def show(name, rows):
    r = d[rows].astype(np.float64)
    print(f"{name:8s} n={len(rows):3d}  s1 {r[:,0].mean():8.0f}  s2 {r[:,1].mean():8.0f}  s3 {r[:,2].mean():8.0f}  end {r[:,3].mean():8.0f} (max {r[:,3].max():8.0f}) ticks")
show("dW2", list(range(0, 8)))
show("dW1", list(range(8, nb - 1)))
print("stepper: pow done", d[nb-1][0], " all-arrived seen", d[nb-1][1], " end", d[nb-1][3])
r = d[8:nb-1].astype(np.float64)
for k, nm in enumerate(["s1 loads", "s2 arrive", "s3 tile", "end"]):
    print(nm, "pct 10/50/90/99/max", [int(np.percentile(r[:, k], q)) for q in (10, 50, 90, 99, 100)])
slow = np.argsort(r[:, 0])[-8:]
print("slowest s1 blocks (id, s1, end):", [(int(8 + i), int(r[i, 0]), int(r[i, 3])) for i in slow])
# absolute starts per XCD are not comparable, but within one XCD (id % 8) they are: spread of starts inside XCD 0
for xcd in (0, 3):
    ids = [i for i in range(nb) if i % 8 == xcd]
    st = t[ids, 0] - t[ids, 0].min()
    print("XCD", xcd, "start spread ticks (by id):", [int(v) for v in st[:6]], "...", [int(v) for v in st[-4:]], "max", int(st.max()))
