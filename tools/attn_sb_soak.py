#!/usr/bin/env python3
"""Developer soak (GPU box): the one-pass attention backward of csrc/attention_sb.hip launched N times on the same inputs at the C4 shape
(B64 H8 T256 dh64, fused [B,T,3D] layout, ragged key padding) -- every launch must reproduce the first one BIT FOR BIT (the dQ slots are
ordered by contribution counters, not by barriers: a protocol error would show up as a rare different summation order or a hang).
python tools/attn_sb_soak.py [launches]"""
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
import torch  # noqa: E402

from neunet_hip.nn.experimental import attention as A  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
B, H, T = 64, 8, 256
D = H * 64
rng = np.random.default_rng(9)
torch.manual_seed(1)
buf = torch.randn(B, T, 3 * D, device="cuda")
q, k, v = buf[..., 0:D], buf[..., D:2 * D], buf[..., 2 * D:]
kvh = np.ones((B, T), np.int32)
for b in range(B):
    kvh[b, T - int(rng.integers(0, 100)):] = 0
kv = torch.tensor(kvh, device="cuda")
do = torch.randn(B, T, D, device="cuda")
scale = float(np.sqrt(D))
ctx, lse = A.fused_attention_forward(q, k, v, kv, H, scale, True)
gb = torch.empty_like(buf)
outg = (gb[..., 0:D], gb[..., D:2 * D], gb[..., 2 * D:])
A.fused_attention_backward(q, k, v, kv, ctx, lse, H, scale, True, do, out=outg)
ref = gb.clone()
bad = 0
for it in range(n):
    gb.fill_(float("nan"))
    A.fused_attention_backward(q, k, v, kv, ctx, lse, H, scale, True, do, out=outg)
    if it % 50 == 49 or it == n - 1:
        if not torch.equal(gb, ref):
            bad += 1
            print(f"launch {it}: differs from the first", flush=True)
print(f"{n} launches, {bad} mismatching checks -> {'OK' if bad == 0 else 'FAIL'}")
sys.exit(1 if bad else 0)
