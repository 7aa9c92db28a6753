#!/usr/bin/env python3
"""Which host lines of a GPT training step still launch torch's own kernels (fill / copy)?  One eager step at toy size under
torch.profiler with Python stacks; prints the innermost repo frame of every aten::fill_/zero_/copy_/clone/contiguous call."""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch  # noqa: E402
import gpt_tiny  # noqa: E402
import neunet_hip as hip  # noqa: E402
import neunet_hip.nn as nn  # noqa: E402
from neunet_hip.distributed import GradBucket  # noqa: E402
from neunet_hip.optim import Adam  # noqa: E402

V, D, H, F, L, B, T = 1000, 128, 2, 512, 2, 4, 64           # head dim 64: the fused attention path
np.random.seed(0)
model = gpt_tiny.build_gpt(V, D, H, F, L, pad_idx=0, max_len=256, fused=True)
rng = np.random.default_rng(0)
b = rng.integers(3, V, (B, T + 1)).astype(np.int32)
b[1, -9:] = 0
ids = hip.Tensor(np.ascontiguousarray(b[:, :-1]), dtype=np.int32, requires_grad=False, device="cuda")
tgt = hip.Tensor(np.ascontiguousarray(b[:, 1:]).reshape(-1), dtype=np.int32, requires_grad=False, device="cuda")
loss_fn = nn.CrossEntropyLoss(ignore_index=0)


def fb():
    out, _ = model.forward(ids)
    loss = loss_fn(out.reshape(B * T, V), tgt)
    loss.backward()
    return loss


fb()
active = [p for p in model.parameters() if p.grad is not None]
opt = Adam(model.parameters(), lr=1e-4)
opt.zero_grad()
bucket = GradBucket(active)


def step():
    opt.zero_grad()
    fb()
    bucket.collect()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
hits = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::clone", "aten::contiguous", "aten::zeros", "aten::ones_like",
                   "aten::zeros_like", "aten::to", "aten::_to_copy", "aten::ne", "aten::eq"):
        frame = next((f for f in ev.stack if "/root/repo" in f or "neunet_hip" in f or "gpt_tiny" in f or "bench.py" in f), "?")
        hits[(ev.name, frame.strip()[-110:])] += 1
for (name, frame), n in sorted(hits.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d} x {name:18s} {frame}")
