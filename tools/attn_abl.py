"""Developer tool: run the fused attention forward + backward at the C4 shape (B64 T256 H8 d512) N times so that
rocprofv3 (tools/attn_abl.sh, tools/attn_pmc.sh) can collect per-kernel durations / counters.  ATTN_CAUSAL=0 for the
non-causal case."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
import torch  # noqa: E402

from neunet_hip.nn.experimental.attention import fused_attention_backward, fused_attention_forward  # noqa: E402

B, T, H = 64, 256, 8
D = H * 64
q, k, v, do = [torch.randn(B, T, D, device="cuda") for _ in range(4)]
kv = torch.ones(B, T, dtype=torch.int32, device="cuda")
causal = os.environ.get("ATTN_CAUSAL", "1") == "1"
for _ in range(40):
    ctx, lse = fused_attention_forward(q, k, v, kv, H, 22.6, causal)
    fused_attention_backward(q, k, v, kv, ctx, lse, H, 22.6, causal, do)
torch.cuda.synchronize()
