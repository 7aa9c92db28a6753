import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
import torch
from neunet_hip.nn.experimental import attention as A
torch.manual_seed(1)
torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)
T = 256; B, H = 1, 2; D = 128
scale = float(np.sqrt(D))
q, k, v = [torch.randn(B, T, D, device="cuda") for _ in range(3)]
do = torch.randn(B, T, D, device="cuda")
k[0, :, 64:80] = 0.0          # slice 1: K columns 0..15 are zero -> dQ columns 0..15 of slice 1 must be exactly 0
ref, attn, _ = A.attention_forward(q, k, v, None, H, scale, True)
gref = A.attention_backward(q, k, v, attn, None, H, scale, True, do)
out, lse = A.fused_attention_forward(q, k, v, None, H, scale, True)
g = A.fused_attention_backward(q, k, v, None, out, lse, H, scale, True, do)
x = g[0][0, :32, 64:80]
print("dq slice1 rg0 cols 0..15 (expect 0): max", x.abs().max().item(), " ref max", gref[0][0, :32, 64:80].abs().max().item())
print(x[:4])
print("ratio to q cols:", (x[:4] / q[0, :4, 64:80]))
print("ratio to do cols:", (x[:4] / do[0, :4, 64:80]))
e = (g[0] - gref[0]).abs()[0, :, 64:128].reshape(8, 32, 64).amax(dim=1)
print("err per row group x col (max over rows), slice 1:", e[:, :20])
