import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd")); sys.path.insert(0, os.path.join(ROOT, "examples")); sys.path.insert(0, ROOT)
import numpy as np, torch
import neunet_hip, neunet_hip.nn as nn
from neunet_hip import Tensor
from neunet_hip.distributed import GradBucket
from neunet_hip.graph import GraphedTrainStep
from neunet_hip.optim import Adam
import gpt_tiny
B, T, D, H, V, L = [int(os.environ.get(k, d)) for k, d in (("DB", 4), ("DT", 32), ("DD", 64), ("DH", 4), ("DV", 101), ("DL", 2))]
rng = np.random.default_rng(0)
model = gpt_tiny.build_gpt(V, D, H, 4*D, L, pad_idx=0, max_len=1024)
ids = Tensor(rng.integers(1, V, (B, T)), dtype=np.int32, requires_grad=False, device="cuda")
tgt = Tensor(rng.integers(1, V, B*T), dtype=np.int32, requires_grad=False, device="cuda")
lf = nn.CrossEntropyLoss(ignore_index=0)
params = model.parameters()
opt = Adam(params, lr=1e-4)
def fb():
    out, _ = model.forward(ids); l = lf(out.reshape(B*T, V), tgt); l.backward(); return l
fb(); active = [p for p in params if p.grad is not None]; opt.zero_grad()
bucket = GradBucket(active, extra_scalars=1)
print("bucket", bucket.numel, len(active), flush=True)
g = GraphedTrainStep(fb, opt, bucket, warmup=2)
print("captured", flush=True)
mode = os.environ.get("MODE", "both")
for i in range(3):
    if mode in ("both", "fb"):
        g.g_fb.replay(); torch.cuda.synchronize(); print("fb replay", i, g.loss.item(), flush=True)
    if mode in ("both", "opt"):
        g.g_opt.replay(); torch.cuda.synchronize(); print("opt replay", i, float(params[0].data.abs().sum()), flush=True)
