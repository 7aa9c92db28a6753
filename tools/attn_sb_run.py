"""Developer tool: N forward + backward launches of the fused attention at the C4 shape (B64 T256 H8 dh64, causal, fused [B,T,3D] layout)
for rocprofv3 (tools/attn_sb_pmc.sh).  NNHIP_ATTN_SB=0 runs the tiled kernels instead."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
import torch  # noqa: E402

from neunet_hip.nn.experimental.attention import fused_attention_backward, fused_attention_forward  # noqa: E402

B, T, H = 64, 256, 8
D = H * 64
buf = torch.randn(B, T, 3 * D, device="cuda")
q, k, v = buf[..., 0:D], buf[..., D:2 * D], buf[..., 2 * D:]
gb = torch.empty_like(buf)
outg = (gb[..., 0:D], gb[..., D:2 * D], gb[..., 2 * D:])
do = torch.randn(B, T, D, device="cuda")
kv = torch.ones(B, T, dtype=torch.int32, device="cuda")
n = int(os.environ.get("ATTN_REPS", "30"))
for _ in range(n):
    ctx, lse = fused_attention_forward(q, k, v, kv, H, 22.6, True)
    fused_attention_backward(q, k, v, kv, ctx, lse, H, 22.6, True, do, out=outg)
torch.cuda.synchronize()
