import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, numpy as np
from neunet_hip.nn.experimental.attention import attention_forward, fused_attention_forward, attention_backward, fused_attention_backward
from kbench import bench
torch.manual_seed(0)
for (B, T, H, causal, pad) in [(2, 256, 8, True, True), (3, 100, 4, True, True), (2, 70, 2, False, True), (1, 300, 2, True, False), (64, 256, 8, True, True)]:
    D = H * 64
    q, k, v = [torch.randn(B, T, D, device="cuda") for _ in range(3)]
    kv = torch.ones(B, T, dtype=torch.int32, device="cuda")
    if pad: kv[0, -T // 5:] = 0
    scale = float(np.sqrt(D))
    ref, attn, _ = attention_forward(q, k, v, kv, H, scale, causal)
    out, lse = fused_attention_forward(q, k, v, kv, H, scale, causal)
    err = (out - ref).abs().max().item()
    print(f"B{B} T{T} H{H} causal{causal}: max|fused-unfused| = {err:.3e}", flush=True)
# fully-masked rows: key 0 padded -> query 0 has no valid key -> uniform over ALL keys
B, T, H = 1, 128, 1; D = 64
q, k, v = [torch.randn(B, T, D, device="cuda") for _ in range(3)]
kv = torch.ones(B, T, dtype=torch.int32, device="cuda"); kv[0, :3] = 0
ref, _, _ = attention_forward(q, k, v, kv, H, 8.0, True)
out, _ = fused_attention_forward(q, k, v, kv, H, 8.0, True)
print("fully-masked rows: ", (out - ref).abs().max().item())
B, T, H = 64, 256, 8; D = 512
q, k, v = [torch.randn(B, T, D, device="cuda") for _ in range(3)]
kv = torch.ones(B, T, dtype=torch.int32, device="cuda")
print("unfused fwd ms", bench(lambda: attention_forward(q, k, v, kv, H, 22.6, True), 20))
print("fused   fwd ms", bench(lambda: fused_attention_forward(q, k, v, kv, H, 22.6, True), 20))
do = torch.randn_like(q)
ctx_u, attn, _ = attention_forward(q, k, v, kv, H, 22.6, True)
ctx_f, lse = fused_attention_forward(q, k, v, kv, H, 22.6, True)
gu = attention_backward(q, k, v, attn, kv, H, 22.6, True, do)
gf = fused_attention_backward(q, k, v, kv, ctx_f, lse, H, 22.6, True, do)
for a, b, n in zip(gf, gu, "qkv"):
    print(f"d{n}: max|fused-unfused| = {(a - b).abs().max().item():.3e}  (max |ref| {b.abs().max().item():.3e})")
print("unfused bwd ms", bench(lambda: attention_backward(q, k, v, attn, kv, H, 22.6, True, do), 20))
print("fused   bwd ms", bench(lambda: fused_attention_backward(q, k, v, kv, ctx_f, lse, H, 22.6, True, do), 20))
