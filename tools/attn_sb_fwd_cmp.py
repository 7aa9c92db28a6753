"""Developer tool: forward output of the balanced attention kernels at small B, H (NNHIP_ATTN_SB_FWD = stream | lds picks the forward);
`save <file>` writes ctx / lse, `cmp <a> <b>` prints where two runs differ, per (batch, head, row group)."""
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))

if sys.argv[1] == "save":
    import torch
    from neunet_hip.nn.experimental import attention as A
    B, H, T = int(os.environ.get("CMP_B", "2")), int(os.environ.get("CMP_H", "2")), 256
    D = H * 64
    g = torch.Generator(device="cuda").manual_seed(3)
    buf = torch.randn(B, T, 3 * D, device="cuda", generator=g)
    q, k, v = buf[..., 0:D], buf[..., D:2 * D], buf[..., 2 * D:]
    kv = torch.ones(B, T, dtype=torch.int32, device="cuda")
    if os.environ.get("CMP_PAD", "0") == "1":
        kv[0, 200:] = 0
    ctx, lse = A.fused_attention_forward(q, k, v, kv, H, float(np.sqrt(D)), True)
    torch.cuda.synchronize()
    np.savez(sys.argv[2], ctx=ctx.cpu().numpy(), lse=lse.cpu().numpy())
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ca, cb = a["ctx"], b["ctx"]
    B, T, D = ca.shape
    H = D // 64
    err = np.abs(ca - cb).reshape(B, 8, 32, H, 64).max(axis=(2, 4))        # [B, row group, H]
    print("max |diff| per (batch, row group, head):")
    for bb in range(B):
        for hh in range(H):
            print(f"b{bb} h{hh}:", " ".join(f"{err[bb, rg, hh]:9.2e}" for rg in range(8)))
    la, lb = a["lse"], b["lse"]
    print("lse max diff", np.abs(la - lb).max(), "ctx max diff", np.abs(ca - cb).max(), "ctx scale", np.abs(ca).max())
