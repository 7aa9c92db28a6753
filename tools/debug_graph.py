import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd")); sys.path.insert(0, os.path.join(ROOT, "examples")); sys.path.insert(0, ROOT)
import numpy as np, torch
import neunet_hip, neunet_hip.nn as nn
from neunet_hip import Tensor

def try_graph(name, make):
    try:
        fn = make()
        if os.environ.get("DUMMY"):
            fn(); torch.cuda.synchronize()
            dummy = torch.zeros(int(os.environ["DUMMY"]), device="cuda")
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): fn()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay(); torch.cuda.synchronize()
            if os.environ.get("ITEM"): print("   item", float(out.data.reshape(-1)[0].item()), flush=True)
        print(f"{name:28s} OK", flush=True)
    except Exception as e:
        print(f"{name:28s} FAIL {type(e).__name__}: {str(e)[:200]}", flush=True)
        torch.cuda.synchronize()

B, T, D, H, V = [int(os.environ.get(k, d)) for k, d in (("DB", 4), ("DT", 32), ("DD", 64), ("DH", 4), ("DV", 101))]
NL = int(os.environ.get("DL", 2))
rng = np.random.default_rng(0)
def mk_linear():
    l = nn.Linear(D, D); x = Tensor(rng.standard_normal((B*T, D)), device="cuda")
    def f():
        x.grad=None; l.weight.grad=None; l.bias.grad=None
        y = l(x); y.backward(); return y
    return f
def mk_linear_splitk():
    l = nn.Linear(64, 64); x = Tensor(rng.standard_normal((4096, 64)), device="cuda")
    def f():
        x.grad=None; l.weight.grad=None; l.bias.grad=None
        y = l(x); y.backward(); return y
    return f
def mk_rms():
    l = nn.RMSNorm(D); x = Tensor(rng.standard_normal((B, T, D)), device="cuda")
    def f():
        x.grad=None; l.weight.grad=None
        y = l(x); y.backward(); return y
    return f
def mk_lswish():
    l = nn.LinearSwish(D, 2*D); x = Tensor(rng.standard_normal((B, T, D)), device="cuda")
    def f():
        x.grad=None; l.weight.grad=None; l.bias.grad=None
        y = l(x); y.backward(); return y
    return f
def mk_emb():
    e = nn.Embedding(V, D); pe = nn.PositionalEncoding(D, 64)
    ids = Tensor(rng.integers(0, V, (B, T)), dtype=np.int32, requires_grad=False, device="cuda")
    def f():
        e.weight.grad=None
        y = e(ids, scale=8.0, pe=pe.table); y.backward(); return y
    return f
def mk_mha():
    m = nn.MultiHeadAttention(D, H); x = Tensor(rng.standard_normal((B, T, D)), device="cuda")
    kv = torch.ones(B, T, dtype=torch.int32, device="cuda")
    def f():
        x.grad=None
        for p in m.parameters(): p.grad=None
        y, _ = m(x, x, x, kv, True); y.backward(); return y
    return f
def mk_ce():
    x = Tensor(rng.standard_normal((B*T, V)), device="cuda")
    y = Tensor(rng.integers(0, V, B*T), dtype=np.int32, requires_grad=False, device="cuda")
    lf = nn.CrossEntropyLoss(ignore_index=0)
    def f():
        x.grad=None
        l = lf(x, y); l.backward(); return l
    return f
def mk_add():
    a = Tensor(rng.standard_normal((B, T, D)), device="cuda"); b = Tensor(rng.standard_normal((B, T, D)), device="cuda")
    def f():
        a.grad=None; b.grad=None
        y = a + b; y.backward(); return y
    return f
def mk_gpt():
    import gpt_tiny
    model = gpt_tiny.build_gpt(V, D, H, 4*D, NL, pad_idx=0, max_len=1024)
    ids = Tensor(rng.integers(1, V, (B, T)), dtype=np.int32, requires_grad=False, device="cuda")
    tgt = Tensor(rng.integers(1, V, B*T), dtype=np.int32, requires_grad=False, device="cuda")
    lf = nn.CrossEntropyLoss(ignore_index=0)
    def f():
        for p in model.parameters(): p.grad=None
        out, _ = model.forward(ids); l = lf(out.reshape(B*T, V), tgt); l.backward(); return l
    return f
for name, mk in [("linear", mk_linear), ("linear_splitk", mk_linear_splitk), ("rmsnorm", mk_rms), ("linear_swish", mk_lswish),
                 ("embedding", mk_emb), ("mha", mk_mha), ("ce", mk_ce), ("add", mk_add), ("gpt", mk_gpt)]:
    if len(sys.argv) > 1 and name not in sys.argv[1:]: continue
    try_graph(name, mk)
