import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd")); sys.path.insert(0, os.path.join(ROOT, "examples")); sys.path.insert(0, ROOT)
import numpy as np, torch
import neunet_hip, neunet_hip.nn as nn
from neunet_hip import Tensor
from neunet_hip.distributed import GradBucket
import gpt_tiny
B, T, D, H, V, L = 4, 32, 64, 4, 101, 2
VAR = os.environ.get("VAR", "C")
rng = np.random.default_rng(0)
model = gpt_tiny.build_gpt(V, D, H, 4*D, L, pad_idx=0, max_len=1024)
ids = Tensor(rng.integers(1, V, (B, T)), dtype=np.int32, requires_grad=False, device="cuda")
tgt = Tensor(rng.integers(1, V, B*T), dtype=np.int32, requires_grad=False, device="cuda")
lf = nn.CrossEntropyLoss(ignore_index=0)
params = model.parameters()
def fb():
    out, _ = model.forward(ids); l = lf(out.reshape(B*T, V), tgt); l.backward(); return l
def zero():
    for p in params: p.grad = None
fb(); active = [p for p in params if p.grad is not None]; zero()
bucket = None
if VAR == "E":
    dummy = torch.zeros(112748, device="cuda")
if VAR == "F":
    dummy = [torch.zeros(int(np.prod(p.shape)), device="cuda") for p in active]
if VAR == "G":   # slots attached by hand to separately allocated tensors (no flat bucket)
    for p in active:
        p._grad_slot = torch.zeros(tuple(p.shape), device="cuda")
if VAR in ("B", "C", "D"):
    bucket = GradBucket(active)
    if VAR == "B": bucket.detach()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        zero(); fb()
        if bucket and VAR != "D": bucket.collect()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); zero()
with torch.cuda.graph(g):
    loss = fb()
    if bucket and VAR != "D": bucket.collect()
torch.cuda.synchronize(); print(VAR, "captured", flush=True)
for i in range(3):
    g.replay(); torch.cuda.synchronize(); print(VAR, "replay", i, loss.item() if not os.environ.get("NOITEM") else "-", flush=True)
