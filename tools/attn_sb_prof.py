#!/usr/bin/env python3
"""Developer tool: where do the cycles of the balanced attention kernels (csrc/attention_sb.hip) go?  Needs the instrumented build
(python numpy-nn-model_amd/build.py --variant prof -D SB_PROF).  Runs the C4 shape once with the profile buffer on and prints the mean
clock64 ticks per phase and wave."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "numpy-nn-model_amd", "neunet_hip", "lib", "libneunet_hip.prof.so")
os.environ["NEUNET_HIP_LIB"] = LIB
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
import torch  # noqa: E402

from neunet_hip.nn.experimental import attention as A  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B, T, H = 64, 256, 8
D = H * 64
scale = float(np.sqrt(D))
buf = torch.randn(B, T, 3 * D, device="cuda")
q, k, v = buf[..., 0:D], buf[..., D:2 * D], buf[..., 2 * D:]
gb = torch.empty_like(buf)
outg = (gb[..., 0:D], gb[..., D:2 * D], gb[..., 2 * D:])
kv = torch.ones(B, T, dtype=torch.int32, device="cuda")
do = torch.randn(B, T, D, device="cuda")
dll = ctypes.CDLL(LIB)
dll.nnhipAttentionSbSetProfile.argtypes = [ctypes.c_void_p]
nblk = B * H // 2
prof = torch.zeros(nblk * 8 * 16, dtype=torch.int64, device="cuda")
for _ in range(3):
    ctx, lse = A.fused_attention_forward(q, k, v, kv, H, scale, True)
    A.fused_attention_backward(q, k, v, kv, ctx, lse, H, scale, True, do, out=outg)
torch.cuda.synchronize()
assert dll.nnhipAttentionSbSetProfile(ctypes.c_void_p(prof.data_ptr())) == 0
if which == "fwd":
    A.fused_attention_forward(q, k, v, kv, H, scale, True)
    names = ["prologue", "unit top", "QK MFMAs", "softmax", "PV MFMAs", "loop glue", "epilogue"]
else:
    A.fused_attention_backward(q, k, v, kv, ctx, lse, H, scale, True, do, out=outg)
    names = ["setup", "U1 S", "U2 dP", "P / dS", "U3 dV", "U4 dK (+dS^T write)", "dS^T read", "U5 dQ", "slot", "barrier", "epilogue", "glue"]
torch.cuda.synchronize()
dll.nnhipAttentionSbSetProfile(None)
pr = prof.cpu().numpy().reshape(nblk, 8, 16).astype(np.float64)
tot = pr.sum(axis=2)
print(f"{which}: mean ticks per wave {tot.mean():.0f} (min {tot.min():.0f}, max {tot.max():.0f})")
for w in range(8):
    print(f"  wave {w}: total {tot[:, w].mean():8.0f}  " + "  ".join(f"{names[i]} {pr[:, w, i].mean():7.0f}" for i in range(len(names))))
print("  all    : " + "  ".join(f"{names[i]} {pr[:, :, i].mean() / tot.mean() * 100:5.1f}%" for i in range(len(names))))
