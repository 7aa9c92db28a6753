"""Developer tool: does a 128-byte-aligned row pitch of the logits (vocabulary 15000 -> leading dimension 15008) pay?
The three GEMMs of the C4 vocabulary head (16384 x 512 -> 15000) through nnhipGemmF32Ex and the fused CrossEntropy, with the
logits / d(logits) buffer at row pitch 15000 and 15008 floats."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from neunet_hip import _lib  # noqa: E402
from neunet_hip._lib import call_hip_function as call  # noqa: E402
from kbench import bench  # noqa: E402

st = _lib.get_current_stream_ptr()
M, K, V = 16384, 512, 15000
X = torch.randn(M, K, device="cuda")
W = torch.randn(V, K, device="cuda") / 22
b = torch.randn(V, device="cuda")
dX = torch.empty(M, K, device="cuda")
dW = torch.empty(V, K, device="cuda")
labels = torch.randint(0, V, (M,), device="cuda", dtype=torch.int32)
loss_rows, lse = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
loss, cnt = torch.empty(1, device="cuda"), torch.empty(1, device="cuda", dtype=torch.int32)
fl = 2.0 * M * K * V
for ld in (15000, 15008, 15040, 15104):
    buf = torch.randn(M, ld, device="cuda")
    L = buf[:, :V]
    f = bench(lambda: call("nnhipGemmF32Ex", X, W, _lib.StridedView(L), b, M, V, K, K, K, ld, 1, 1, 1, 0, 0, 0, 1, 0, 0, 0, 1.0, st), 20)
    ce = bench(lambda: call("nnhipCrossEntropyLossEx", _lib.StridedView(L), None, loss_rows, lse, labels, 4, None, ld, -100, M, V, b"m", loss, cnt, st), 20)
    gx = bench(lambda: call("nnhipGemmF32Ex", _lib.StridedView(L), W, dX, None, M, K, V, ld, K, K, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1.0, st), 20)
    gw = bench(lambda: call("nnhipGemmF32Ex", _lib.StridedView(L), X, dW, None, V, K, M, ld, K, K, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1.0, st), 20)
    print(f"ld {ld}: fwd {f[0] * 1e3:7.1f} us ({fl / f[0] / 1e9:5.1f} TF)  CE {ce[0] * 1e3:6.1f} us  dX {gx[0] * 1e3:7.1f} us ({fl / gx[0] / 1e9:5.1f} TF)  "
          f"dW {gw[0] * 1e3:7.1f} us ({fl / gw[0] / 1e9:5.1f} TF)  sum {(f[0] + ce[0] + gx[0] + gw[0]) * 1e3:7.1f} us", flush=True)
