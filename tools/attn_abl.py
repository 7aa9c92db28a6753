import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, numpy as np
from neunet_hip.nn.experimental.attention import fused_attention_forward
from kbench import bench
B, T, H = 64, 256, 8; D = 512
q, k, v = [torch.randn(B, T, D, device="cuda") for _ in range(3)]
kv = torch.ones(B, T, dtype=torch.int32, device="cuda")
causal = os.environ.get("ATTN_CAUSAL", "1") == "1"
from neunet_hip.nn.experimental.attention import fused_attention_backward
do = torch.randn_like(q)
for _ in range(40):
    ctx, lse = fused_attention_forward(q, k, v, kv, H, 22.6, causal)
    fused_attention_backward(q, k, v, kv, ctx, lse, H, 22.6, causal, do)
torch.cuda.synchronize()
sys.exit(0)
print("ABL", os.environ.get("NNHIP_ATTN_ABL"), "causal", bench(lambda: fused_attention_forward(q, k, v, kv, H, 22.6, True), 30),
      "full", bench(lambda: fused_attention_forward(q, k, v, kv, H, 22.6, False), 30))
