"""Developer tool: time the fused attention forward / backward at the C4 shape with whatever library NEUNET_HIP_LIB names."""
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from neunet_hip.nn.experimental import attention as A  # noqa: E402
from kbench import bench  # noqa: E402

B, T, H = 64, 256, 8
D = H * 64
scale = float(np.sqrt(D))
buf = torch.randn(B, T, 3 * D, device="cuda")
q, k, v = buf[..., 0:D], buf[..., D:2 * D], buf[..., 2 * D:]
gb = torch.empty_like(buf)
outg = (gb[..., 0:D], gb[..., D:2 * D], gb[..., 2 * D:])
kv = torch.ones(B, T, dtype=torch.int32, device="cuda")
do = torch.randn(B, T, D, device="cuda")
ctx, lse = A.fused_attention_forward(q, k, v, kv, H, scale, True)
f = bench(lambda: A.fused_attention_forward(q, k, v, kv, H, scale, True), 40)
b = bench(lambda: A.fused_attention_backward(q, k, v, kv, ctx, lse, H, scale, True, do, out=outg), 40)
print(f"{os.path.basename(os.environ.get('NEUNET_HIP_LIB', 'default')):40s} fwd med {f[0] * 1e3:6.1f} min {f[1] * 1e3:6.1f} us | bwd med {b[0] * 1e3:6.1f} min {b[1] * 1e3:6.1f} us", flush=True)
