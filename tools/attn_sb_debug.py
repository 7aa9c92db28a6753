import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
import torch
from neunet_hip.nn.experimental import attention as A
torch.manual_seed(1)
T = 256
for (B, H) in [(1, 2), (2, 1), (3, 1), (1, 4)]:
    D = H * 64
    scale = float(np.sqrt(D))
    q, k, v = [torch.randn(B, T, D, device="cuda") for _ in range(3)]
    do = torch.randn(B, T, D, device="cuda")
    ref, attn, _ = A.attention_forward(q, k, v, None, H, scale, True)
    gref = A.attention_backward(q, k, v, attn, None, H, scale, True, do)
    os.environ["NNHIP_ATTN_SB"] = "1"
    out, lse = A.fused_attention_forward(q, k, v, None, H, scale, True)
    g = A.fused_attention_backward(q, k, v, None, out, lse, H, scale, True, do)
    for name, a, b in zip("qkv", g, gref):
        e = (a - b).abs().reshape(B, 8, 32, H, 64).amax(dim=(2, 4))   # [B, row group, H]
        print(f"B{B} H{H} d{name} max|ref| {b.abs().max().item():.3f}")
        for bb in range(B):
            for hh in range(H):
                print(f"   slice bh={bb * H + hh} (phase {(bb * H + hh) % 2}): " + " ".join(f"{x:.1e}" for x in e[bb, :, hh].tolist()))
B, H = 1, 2
D = H * 64
scale = float(np.sqrt(D))
q, k, v = [torch.randn(B, T, D, device="cuda") for _ in range(3)]
do = torch.randn(B, T, D, device="cuda")
ref, attn, _ = A.attention_forward(q, k, v, None, H, scale, True)
gref = A.attention_backward(q, k, v, attn, None, H, scale, True, do)
out, lse = A.fused_attention_forward(q, k, v, None, H, scale, True)
g = A.fused_attention_backward(q, k, v, None, out, lse, H, scale, True, do)
e = (g[0] - gref[0]).abs()[0, :32, 64:128]      # slice 1, row group 0: [32 rows, 64 cols]
torch.set_printoptions(linewidth=250, precision=1, sci_mode=True)
print("per-row max err:", e.amax(dim=1))
print("per-col max err:", e.amax(dim=0))
print("got row 5:", g[0][0, 5, 64:80])
print("ref row 5:", gref[0][0, 5, 64:80])
print("got row 20:", g[0][0, 20, 64:80])
print("ref row 20:", gref[0][0, 20, 64:80])
