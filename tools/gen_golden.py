#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (imported from /root/reference).

Run in the build container only (the reference never travels to the GPU box):

    python tools/gen_golden.py            # writes tests/golden/*.npz

The reference does `import cupy` unconditionally (neunet/autograd.py:3), so a stub
package from tools/oracle_stub/ is put on sys.path first.  Every fixture stores explicit
input arrays AND the reference's outputs/gradients; tests never re-derive inputs from
RNG state.  Fixtures are data only -- no reference source is copied.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "oracle_stub"))
sys.path.insert(0, "/root/reference")

import neunet  # noqa: E402
import neunet.nn as nn  # noqa: E402
from neunet.autograd import Tensor  # noqa: E402
from neunet.optim import Adam, AdamW  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
F32 = np.float32
QUIET = False


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    if not QUIET:
        print(f"{name:28s} {os.path.getsize(path) / 1024:8.1f} KiB")


def seed_layers(n):
    """The reference layers draw their initial weights from the GLOBAL np.random (nn.Linear / Conv2d / Embedding
    __init__): seed it at the top of every generator so every fixture regenerates bit for bit
    (tests/test_oracle_golden.py::test_fixtures_regenerate_bit_for_bit)."""
    np.random.seed(n)


def T(a, **kw):
    return Tensor(a, **kw)


def set_linear(layer, W, b):
    layer.weight.data[...] = W
    if b is not None:
        layer.bias.data[...] = b


# --------------------------------------------------------------------------- Linear
def gen_linear():
    seed_layers(100)
    rng = np.random.default_rng(11)
    for name, xshape, bias in [("linear_2d", (16, 24), True), ("linear_3d", (4, 6, 24), True),
                               ("linear_nobias", (16, 24), False)]:
        X = rng.uniform(-1, 1, xshape).astype(F32)
        W = rng.uniform(-0.2, 0.2, (40, 24)).astype(F32)
        b = rng.uniform(-0.2, 0.2, (1, 40)).astype(F32) if bias else None
        dO = rng.uniform(-1, 1, xshape[:-1] + (40,)).astype(F32)
        layer = nn.Linear(24, 40, bias=bias)
        set_linear(layer, W, b)
        x = T(X)
        out = layer(x)
        out.backward(dO)
        arrs = dict(X=X, W=W, dO=dO, O=out.data, dX=x.grad, dW=layer.weight.grad)
        if bias:
            arrs.update(b=b, db=layer.bias.grad)
        save(name, **arrs)


# --------------------------------------------------------------------------- activations
def gen_activations():
    seed_layers(101)
    rng = np.random.default_rng(12)
    X = (rng.standard_normal((8, 64)) * 2).astype(F32)
    dY = rng.standard_normal((8, 64)).astype(F32)
    x = T(X)
    y = nn.ReLU()(x)
    y.backward(dY)
    save("relu", X=X, dY=dY, Y=y.data, dX=x.grad)

    for beta in (1.0, 1.5):
        x = T(X)
        y = nn.Swish(beta)(x)
        y.backward(dY)
        save(f"swish_b{beta}", X=X, dY=dY, Y=y.data, dX=x.grad, beta=np.float64(beta))

    # SwiGLU gate: the reference has no CPU class; compose it on the reference tape.
    for tag, shape, h, beta in [("swiglu_2d", (8, 64), 32, 1.0), ("swiglu_3d", (2, 5, 48), 24, 1.5)]:
        Xg = rng.standard_normal(shape).astype(F32)
        dYg = rng.standard_normal(shape[:-1] + (h,)).astype(F32)
        x = T(Xg)
        gate = x[..., :h]
        up = x[..., h:]
        y = nn.Swish(beta)(gate) * up
        y.backward(dYg)
        save(tag, X=Xg, dY=dYg, Y=y.data, dX=x.grad, beta=np.float64(beta))

    for tag, shape, axis in [("softmax_last", (8, 64), -1), ("softmax_axis1_4d", (2, 5, 3, 4), 1),
                             ("softmax_axis1_2d", (6, 50), 1)]:
        Xs = (rng.standard_normal(shape) * 3).astype(F32)
        dYs = rng.standard_normal(shape).astype(F32)
        x = T(Xs)
        y = nn.Softmax(axis=axis)(x)
        y.backward(dYs)
        save(tag, X=Xs, dY=dYs, Y=y.data, dX=x.grad, axis=np.int64(axis))


# --------------------------------------------------------------------------- CrossEntropy
def gen_ce():
    seed_layers(102)
    rng = np.random.default_rng(13)
    cases = [("ce_mean", (16, 128), -100, "mean", 0), ("ce_sum", (16, 128), -100, "sum", 0),
             ("ce_none", (16, 128), -100, "none", 0),
             ("ce_mean_ign", (16, 128), -100, "mean", 5), ("ce_sum_ign", (16, 128), -100, "sum", 5),
             ("ce_none_ign", (16, 128), -100, "none", 5),
             ("ce_mean_pad0", (24, 40), 0, "mean", 7), ("ce_mean_small", (32, 10), -1000, "mean", 0)]
    for tag, (rows, C), ign, red, n_ign in cases:
        logits = (rng.standard_normal((rows, C)) * 2).astype(F32)
        lo = 1 if ign == 0 else 0
        labels = rng.integers(lo, C, rows).astype(np.int32)
        if n_ign:
            labels[rng.choice(rows, n_ign, replace=False)] = ign
        if tag == "ce_mean_small":
            # ignore_index out of python-negative-index range would raise only if present; none are.
            pass
        x = T(logits)
        loss_fn = nn.CrossEntropyLoss(ignore_index=ign, reduction=red)
        loss = loss_fn(x, T(labels, dtype=np.int32, requires_grad=False))
        loss.backward()
        save(tag, logits=logits, labels=labels, loss=np.asarray(loss.data), dlogits=x.grad,
             ignore_index=np.int64(ign), reduction=np.array(red))


def gen_ce_weighted():
    """CrossEntropyLoss(weight=...) (neunet/nn/losses.py:93-118) with ignored labels, the three reductions, int64 labels."""
    seed_layers(150)
    rng = np.random.default_rng(33)
    rows, C = 24, 130                      # C >= 100 so that weight[-100] indexes (losses.py:115 quirk)
    logits = (rng.standard_normal((rows, C)) * 2).astype(F32)
    labels = rng.integers(0, C, rows).astype(np.int64)
    labels[::5] = -100
    w = rng.uniform(0.2, 3.0, C).astype(F32)
    arrs = dict(logits=logits, labels=labels, weight=w, ignore_index=np.int64(-100))
    for red in ("mean", "sum", "none"):
        x = T(logits)
        loss = nn.CrossEntropyLoss(weight=w.copy(), ignore_index=-100, reduction=red)(x, Tensor(labels, dtype=np.int64, requires_grad=False))
        loss.backward()
        arrs[f"loss_{red}"] = np.asarray(loss.data, dtype=F32).reshape(-1)
        arrs[f"dlogits_{red}"] = x.grad
    save("ce_weighted", **arrs)


# --------------------------------------------------------------------------- RMSNorm
def gen_rmsnorm():
    seed_layers(103)
    rng = np.random.default_rng(14)
    for tag, shape, bias in [("rmsnorm_2d", (8, 64), False), ("rmsnorm_3d_bias", (2, 4, 64), True)]:
        X = rng.standard_normal(shape).astype(F32)
        w = rng.uniform(0.5, 1.5, shape[-1]).astype(F32)
        b = rng.uniform(-0.5, 0.5, shape[-1]).astype(F32) if bias else None
        dY = rng.standard_normal(shape).astype(F32)
        layer = nn.RMSNorm(shape[-1], eps=1e-6, bias=bias)
        layer.weight.data[...] = w
        if bias:
            layer.bias.data[...] = b
        x = T(X)
        y = layer(x)
        y.backward(dY)
        arrs = dict(X=X, w=w, dY=dY, Y=y.data, dX=x.grad, dw=layer.weight.grad, eps=np.float64(1e-6))
        if bias:
            arrs.update(b=b, db=layer.bias.grad)
        save(tag, **arrs)


# --------------------------------------------------------------------------- Conv2d
def gen_conv():
    seed_layers(104)
    rng = np.random.default_rng(15)
    cases = [
        ("conv2d_s2p1d2", (2, 3, 9, 9), 4, 3, (2, 2), (1, 1), (2, 2)),
        ("conv2d_s2_uncovered", (2, 2, 8, 7), 3, (3, 2), (2, 3), (0, 1), (1, 1)),
        # string paddings ("same"/"valid") are unreachable in the reference: __init__ wraps a str
        # into a 2-tuple (conv2d.py:164) so build() never sees the bare string -> TypeError.
        ("conv2d_pad4", (1, 2, 6, 6), 3, 3, (1, 1), (1, 2, 0, 1), (1, 1)),
        ("conv2d_c5_l1", (2, 1, 28, 28), 8, 3, (1, 1), (1, 1), (1, 1)),
        ("conv2d_c5_l2", (2, 8, 14, 14), 16, 3, (1, 1), (1, 1), (1, 1)),
    ]
    for tag, xshape, cout, ks, stride, pad, dil in cases:
        X = rng.uniform(-1, 1, xshape).astype(F32)
        layer = nn.Conv2d(xshape[1], cout, ks, stride, pad, dil)
        W = layer.weight.data.copy()
        b = rng.uniform(-0.3, 0.3, cout).astype(F32)
        layer.bias.data[...] = b
        x = T(X)
        y = layer(x)
        dO = rng.uniform(-1, 1, y.shape).astype(F32)
        y.backward(dO)
        pad_arr = np.array(pad)
        save(tag, X=X, W=W, b=b, dO=dO, O=y.data, dX=x.grad, dW=layer.weight.grad, db=layer.bias.grad,
             stride=np.array(stride), padding=pad_arr, dilation=np.array(dil),
             padding4=np.array(layer.padding))


# --------------------------------------------------------------------------- Adam / AdamW
def gen_adam():
    seed_layers(105)
    rng = np.random.default_rng(16)
    shapes = [(8, 16), (1, 16), (5,)]
    for tag, cls, wd in [("adam_wd0", Adam, 0.0), ("adam_wd1e-2", Adam, 1e-2),
                         ("adamw_wd0", AdamW, 0.0), ("adamw_wd1e-2", AdamW, 1e-2)]:
        params = [nn.Parameter(T(rng.standard_normal(s).astype(F32))) for s in shapes]
        p0 = [p.data.copy() for p in params]
        opt = cls(params, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
        grads, ps, ms, vs = [], [], [], []
        for step in range(3):
            gs = [rng.standard_normal(s).astype(F32) for s in shapes]
            for p, g in zip(params, gs):
                p.grad = g
            opt.step()
            grads.append(gs)
            ps.append([p.data.copy() for p in params])
            ms.append([m.copy() for m in opt.m])
            vs.append([v.copy() for v in opt.v])
        arrs = {"wd": np.float64(wd), "lr": np.float64(1e-2), "n_tensors": np.int64(len(shapes))}
        for i in range(len(shapes)):
            arrs[f"p0_{i}"] = p0[i]
            for s in range(3):
                arrs[f"g{s}_{i}"] = grads[s][i]
                arrs[f"p{s + 1}_{i}"] = ps[s][i]
                arrs[f"m{s + 1}_{i}"] = ms[s][i]
                arrs[f"v{s + 1}_{i}"] = vs[s][i]
        save(tag, **arrs)


# --------------------------------------------------------------------------- Linear->Swish
def gen_linear_swish():
    seed_layers(106)
    rng = np.random.default_rng(17)
    X = rng.uniform(-1, 1, (16, 24)).astype(F32)
    W = rng.uniform(-0.4, 0.4, (40, 24)).astype(F32)
    b = rng.uniform(-0.4, 0.4, (1, 40)).astype(F32)
    dY = rng.uniform(-1, 1, (16, 40)).astype(F32)
    layer = nn.Linear(24, 40)
    set_linear(layer, W, b)
    x = T(X)
    y = nn.Swish(1.5)(layer(x))
    y.backward(dY)
    save("linear_swish", X=X, W=W, b=b, dY=dY, Y=y.data, dX=x.grad, dW=layer.weight.grad,
         db=layer.bias.grad, beta=np.float64(1.5))


# --------------------------------------------------------------------------- C1 MLP trajectory
def gen_mlp():
    seed_layers(107)
    """README.md:57-71 quick-start loop: 784->128->10, CE(mean), Adam(lr 1e-3), batch 32, 3 steps."""
    rng = np.random.default_rng(1001)

    class MLP(nn.Module):
        def __init__(self):
            self.l1 = nn.Linear(784, 128)
            self.relu = nn.ReLU()
            self.l2 = nn.Linear(128, 10)

        def forward(self, x):
            return self.l2(self.relu(self.l1(x)))

    W1 = rng.uniform(-1 / 28, 1 / 28, (128, 784)).astype(F32)
    b1 = rng.uniform(-1 / 28, 1 / 28, (1, 128)).astype(F32)
    s2 = 1 / np.sqrt(128)
    W2 = rng.uniform(-s2, s2, (10, 128)).astype(F32)
    b2 = rng.uniform(-s2, s2, (1, 10)).astype(F32)
    model = MLP()
    set_linear(model.l1, W1, b1)
    set_linear(model.l2, W2, b2)
    opt = Adam(model.parameters(), lr=1e-3)
    loss_fn = nn.CrossEntropyLoss()
    X = rng.uniform(-1, 1, (3, 32, 784)).astype(F32)
    Y = rng.integers(0, 10, (3, 32)).astype(np.int32)
    losses, argmaxes, first_grads = [], [], None
    for s in range(3):
        opt.zero_grad()
        out = model(T(X[s]))
        loss = loss_fn(out, T(Y[s], dtype=np.int32, requires_grad=False))
        loss.backward()
        if s == 0:
            first_grads = [p.grad.copy() for p in model.parameters()]
        opt.step()
        losses.append(float(loss.data))
        argmaxes.append(np.asarray(neunet.argmax(out, axis=1).data))
    Wf = model.l1.weight.data
    save("mlp_c1", W1=W1, b1=b1, W2=W2, b2=b2, X=X, Y=Y,
         losses=np.array(losses, dtype=np.float64), argmax=np.stack(argmaxes).astype(np.int32),
         W1_final_rows=Wf[::8].copy(), W1_final_sum=np.float64(Wf.astype(np.float64).sum()),
         b1_final=model.l1.bias.data, W2_final=model.l2.weight.data, b2_final=model.l2.bias.data,
         dW1_step0_rows=first_grads[0][::8].copy(), db1_step0=first_grads[1],
         dW2_step0=first_grads[2], db2_step0=first_grads[3])


# --------------------------------------------------------------------------- GPT (examples/gpt.ipynb)
def _notebook_namespace():
    """exec() the notebook's model cells (2-7) straight from /root/reference/examples/gpt.ipynb -- nothing of
    the notebook is copied into this repository."""
    import json
    import math
    from typing import Optional
    nb = json.load(open("/root/reference/examples/gpt.ipynb"))
    ns = {"nn": nn, "neunet": neunet, "Tensor": Tensor, "math": math, "np": np, "Optional": Optional, "device": "cpu"}
    for idx in (2, 3, 4, 5, 6, 7):
        exec("".join(nb["cells"][idx]["source"]), ns)
    return ns


def gen_gpt():
    seed_layers(108)
    ns = _notebook_namespace()
    rng = np.random.default_rng(18)
    # --- embedding with repeated ids (last-write-wins gradient)
    emb = nn.Embedding(11, 8)
    W = emb.weight.data.copy()
    ids = np.array([[1, 4, 4, 2], [4, 0, 1, 1]], dtype=np.int32)
    out = emb(Tensor(ids, dtype=np.int32, requires_grad=False))
    g = rng.standard_normal(out.shape).astype(F32)
    out.backward(g)
    save("embedding", W=W, ids=ids, out=out.data, grad=g, dW=emb.weight.grad)

    # --- multi-head self-attention with pad + causal mask
    B, T, D, H = 2, 8, 32, 4
    mha = ns["MultiHeadAttention"](D, H, dropout=0.0)
    X = rng.standard_normal((B, T, D)).astype(F32)
    tok = rng.integers(1, 20, (B, T))
    tok[1, -3:] = 0
    gpt_helper = ns["GPT"](decoder=None, pad_idx=0)
    mask = (gpt_helper.get_pad_mask(tok) & gpt_helper.get_sub_mask(tok)).astype(np.int32)
    x = Tensor(X)
    y, attn = mha(x, x, x, Tensor(mask, dtype=np.int32, requires_grad=False))
    dY = rng.standard_normal(y.shape).astype(F32)
    y.backward(dY)
    ps = [mha.wq, mha.wk, mha.wv, mha.fc]
    arrs = dict(X=X, mask=mask, key_valid=(tok != 0).astype(np.int32), Y=y.data, attn=attn.data, dY=dY, dX=x.grad,
                n_heads=np.int64(H))
    for name, lin in zip("qkvo", ps):
        arrs[f"W{name}"], arrs[f"b{name}"] = lin.weight.data, lin.bias.data
        arrs[f"dW{name}"], arrs[f"db{name}"] = lin.weight.grad, lin.bias.grad
    save("mha", **arrs)

    # --- GPT-tiny: 2 layers, d=32, 4 heads, d_ff=64, vocab 50, B=2, T=8, dropout 0, one training step
    V, D, H, F, L, B, T = 50, 32, 4, 64, 2, 2, 9
    dec = ns["Decoder"](tgt_vocab_size=V, d_model=D, n_heads=H, d_ff=F, n_layers=L, dropout=0.0, max_len=64)
    model = ns["GPT"](decoder=dec, pad_idx=0)
    batch = rng.integers(3, V, (B, T)).astype(np.int64)
    batch[1, -3:] = 0
    batch[0, 2] = batch[0, 5]                       # repeated token ids inside one batch
    params = model.parameters()
    p0 = [p.data.copy() for p in params]
    # a checkpoint written BY THE REFERENCE (neunet.save = pickle of state_dict(), neunet/__init__.py:26-29,
    # nn/modules.py:76-86): data only (an OrderedDict of NumPy arrays under the reference's key names)
    neunet.save(model.state_dict(), os.path.join(OUT, "gpt_tiny_state.pkl"))
    opt = Adam(params, lr=1.5e-4, betas=(0.9, 0.98), eps=1e-9)
    loss_fn = nn.CrossEntropyLoss(ignore_index=0)
    output, _ = model.forward(batch[:, :-1])
    logits = output.data.copy()
    output = output.reshape(output.shape[0] * output.shape[1], output.shape[2])
    loss = loss_fn(output, neunet.tensor(batch[:, 1:].flatten(), dtype=neunet.int32))
    loss.backward()
    grads = [None if p.grad is None else p.grad.copy() for p in params]
    opt.step()
    arrs = dict(batch=batch, loss=np.float64(loss.data), logits=logits, n_params=np.int64(len(params)),
                cfg=np.array([V, D, H, F, L]))
    for i, (a, g, p) in enumerate(zip(p0, grads, params)):
        arrs[f"p{i}"] = a
        arrs[f"has_grad{i}"] = np.bool_(g is not None)
        if g is not None:
            arrs[f"g{i}"] = g
            arrs[f"p_after{i}"] = p.data
    save("gpt_tiny", **arrs)


# --------------------------------------------------------------------------- MaxPool2d with dilation
def gen_maxpool_dilated():
    """MaxPool2d(kernel, stride, padding, dilation > 1) through the reference class (maxpool2d.py:85-249): square dilated
    windows and symmetric padding only -- the reference's backward cannot run otherwise (:50-57, :63-64)."""
    seed_layers(131)
    rng = np.random.default_rng(31)
    X = (rng.standard_normal((2, 3, 11, 11)) * 2).astype(F32)
    X[1, 2, 4, 4] = X[1, 2, 4, 6] = 9.5             # a tie between two taps of one dilated window -> first tap wins
    arrs = {"X": X}
    for tag, ks, st, pad, dil in [("k2s1p0d2", 2, 1, 0, 2), ("k3s2p2d2", 3, 2, 2, 2), ("k2s2p1d3", 2, 2, 1, 3)]:
        x = T(X)
        y = nn.MaxPool2d(ks, st, pad, dil)(x)
        dY = rng.standard_normal(y.shape).astype(F32)
        y.backward(dY)
        arrs.update({f"{tag}_Y": y.data, f"{tag}_dY": dY, f"{tag}_dX": x.grad, f"{tag}_cfg": np.array([ks, st, pad, dil])})
    save("maxpool_dilated", **arrs)


# --------------------------------------------------------------------------- conv classifier (config 5)
def gen_vision():
    seed_layers(109)
    import json
    rng = np.random.default_rng(19)
    X = (rng.standard_normal((2, 3, 7, 6)) * 2).astype(F32)
    X[0, 0, 0, 0] = X[0, 0, 0, 1] = 3.5            # a tie inside one pooling window -> first maximum wins
    arrs = {"X": X}
    for name, mod in [("leaky", nn.LeakyReLU(0.01)), ("sigmoid", nn.Sigmoid())]:
        x = T(X)
        y = mod(x)
        dY = rng.standard_normal(y.shape).astype(F32)
        y.backward(dY)
        arrs.update({f"{name}_Y": y.data, f"{name}_dY": dY, f"{name}_dX": x.grad})
    for tag, ks, st, pad in [("pool22", 2, 2, 0), ("pool32p1", 3, 2, 1), ("pool21_overlap", 2, 1, 0)]:
        x = T(X)
        y = nn.MaxPool2d(ks, st, pad)(x)
        dY = rng.standard_normal(y.shape).astype(F32)
        y.backward(dY)
        arrs.update({f"{tag}_Y": y.data, f"{tag}_dY": dY, f"{tag}_dX": x.grad, f"{tag}_cfg": np.array([ks, st, pad])})
    for tag, affine in [("bn", True), ("bn_noaffine", False)]:
        bn = nn.BatchNorm2d(3, affine=affine)
        if affine:
            bn.weight.data[...] = rng.uniform(0.5, 1.5, (1, 3))
            bn.bias.data[...] = rng.uniform(-0.5, 0.5, (1, 3))
            arrs[f"{tag}_w"], arrs[f"{tag}_b"] = bn.weight.data.copy(), bn.bias.data.copy()
        x = T(X)
        y = bn(x)
        dY = rng.standard_normal(y.shape).astype(F32)
        y.backward(dY)
        arrs.update({f"{tag}_Y": y.data, f"{tag}_dY": dY, f"{tag}_dX": x.grad, f"{tag}_rm": bn.running_mean.data,
                     f"{tag}_rv": bn.running_var.data})
        if affine:
            arrs.update({f"{tag}_dw": bn.weight.grad, f"{tag}_db": bn.bias.grad})
            bn.eval()
            arrs[f"{tag}_Yeval"] = bn(T(X)).data
    P_, T_ = rng.uniform(0, 1, (5, 10)).astype(F32), rng.uniform(0, 1, (5, 10)).astype(F32)
    p = T(P_)
    loss = nn.MSELoss()(p, T(T_, requires_grad=False))
    loss.backward()
    arrs.update(mse_P=P_, mse_T=T_, mse_loss=np.float64(loss.data), mse_dP=p.grad)
    save("vision_ops", **arrs)

    # full classifier: exec the notebook's own class (cell 2 up to the instantiation), batch 4, 2 Adam steps
    nb = json.load(open("/root/reference/examples/convolutional_digits_classifier.ipynb"))
    src = "".join(nb["cells"][2]["source"]).split("classifier = Conv2dClassifier()")[0].replace('device = "cuda"', 'device = "cpu"')
    ns = {"nn": nn, "nnet": neunet, "np": np}
    exec(src, ns)
    model = ns["Conv2dClassifier"]()
    params = model.parameters()
    p0 = [q.data.copy() for q in params]
    opt = Adam(params, lr=0.001)
    Xb = rng.uniform(-1, 1, (2, 4, 1, 28, 28)).astype(F32)
    lab = rng.integers(0, 10, (2, 4))
    Tb = np.zeros((2, 4, 10), F32)
    for s in range(2):
        Tb[s, np.arange(4), lab[s]] = 1
    losses, outs, grads0 = [], [], None
    for s in range(2):
        opt.zero_grad()
        out = model(T(Xb[s]))
        loss = nn.MSELoss()(out, T(Tb[s], requires_grad=False))
        loss.backward()
        if s == 0:
            grads0 = [q.grad.copy() for q in params]
        opt.step()
        losses.append(float(loss.data))
        outs.append(out.data.copy())
    arrs = dict(X=Xb, T=Tb, losses=np.array(losses), outs=np.stack(outs), n_params=np.int64(len(params)),
                rm=model.bnorm.running_mean.data, rv=model.bnorm.running_var.data)
    for i, (a, g, q) in enumerate(zip(p0, grads0, params)):
        arrs[f"p{i}"], arrs[f"g{i}"], arrs[f"pf{i}"] = a, g, q.data
    save("conv_classifier", **arrs)


GENERATORS = [gen_linear, gen_activations, gen_ce, gen_ce_weighted, gen_rmsnorm, gen_conv, gen_adam, gen_linear_swish, gen_mlp,
              gen_gpt, gen_vision, gen_maxpool_dilated]


def generate_all(out_dir=None, quiet=False):
    global OUT, QUIET
    if out_dir is not None:
        OUT = out_dir
    QUIET = quiet
    for g in GENERATORS:
        g()


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None, help="write the fixtures here instead of tests/golden")
    a = ap.parse_args()
    generate_all(a.out)
