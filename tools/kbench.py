#!/usr/bin/env python3
"""Per-kernel micro-benchmark through the C ABI (HIP events on the launch stream).  Developer tool for A/B
work on the GPU box:   python tools/kbench.py [--only gemm,swish,...] [--iters 30]
Prints one line per op: median / min ms and the algorithmic TFLOP/s or GB/s (SURVEY 8d figures)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))

import torch  # noqa: E402

from neunet_hip import _lib  # noqa: E402
from neunet_hip._lib import Conv2dDesc, call_hip_function as call  # noqa: E402
import ctypes  # noqa: E402


COLD = {"on": False, "scrub": None}


def bench(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if COLD["on"]:            # --cold: 1.5 GiB of unrelated traffic between launches (L2 / MALL hold none of the operands,
            COLD["scrub"].add_(1.0)   # and carry a dirty tail to write back, like the previous kernel of a real step)
        a.record()
        fn()
        b.record()
        ts.append((a, b))
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ts])
    return float(np.median(ms)), float(ms.min())


def report(name, med, mn, flops=None, nbytes=None):
    s = f"{name:34s} med {med:9.4f} ms  min {mn:9.4f} ms"
    if flops:
        s += f"  {flops / (med * 1e-3) / 1e12:8.2f} TFLOP/s ({flops / (med * 1e-3) / 1e12 / 157.3 * 100:5.1f}% of 157.3)"
    if nbytes:
        s += f"  {nbytes / (med * 1e-3) / 1e9:8.1f} GB/s ({nbytes / (med * 1e-3) / 1e9 / 8000 * 100:5.1f}% of 8 TB/s)"
    print(s, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--gemm-mode", type=int, default=0, help="0 = exact fp32 MFMA (default), 1 = split-bf16 (bf16x3)")
    ap.add_argument("--cold", action="store_true", help="scrub the caches between timed launches")
    args = ap.parse_args()
    if args.cold:
        COLD["on"], COLD["scrub"] = True, torch.zeros(3 * (1 << 27), device="cuda")
    call("nnhipSetGemmMode", args.gemm_mode)
    only = set(filter(None, args.only.split(",")))
    want = lambda k: not only or k in only  # noqa: E731
    st = _lib.get_current_stream_ptr()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1)  # noqa: E731
    randn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731

    if want("gemm"):
        for (M, N, K) in [(4096, 4096, 4096), (8192, 4096, 4096), (16384, 512, 512), (16384, 2048, 512),
                          (16384, 512, 2048), (16384, 15000, 512)]:
            X, W, b = rnd(M, K), rnd(N, K) / 64, rnd(N)
            O_, dO, dX, dW, db = torch.empty(M, N, device=dev), rnd(M, N), torch.empty(M, K, device=dev), \
                torch.empty(N, K, device=dev), torch.empty(N, device=dev)
            fl = 2.0 * M * N * K
            tag = f"{M}x{K}->{N}"
            report(f"linear fwd {tag}", *bench(lambda: call("nnhipLinearModuleForward", X, W, b, O_, M, K, N, st), args.iters), flops=fl)
            report(f"linear dX  {tag}", *bench(lambda: call("nnhipLinearModuleBackward", X, W, dO, dX, None, None, M, K, N, st), args.iters), flops=fl)
            report(f"linear dW  {tag}", *bench(lambda: call("nnhipLinearModuleBackward", X, W, dO, None, dW, None, M, K, N, st), args.iters), flops=fl)
            report(f"linear dW+db {tag}", *bench(lambda: call("nnhipLinearModuleBackward", X, W, dO, None, dW, db, M, K, N, st), args.iters), flops=fl)
            report(f"linear dX+dW+db {tag}", *bench(lambda: call("nnhipLinearModuleBackward", X, W, dO, dX, dW, db, M, K, N, st), args.iters), flops=2 * fl)
            report(f"linear db  {tag}", *bench(lambda: call("nnhipLinearModuleBackward", X, W, dO, None, None, db, M, K, N, st), args.iters), nbytes=4.0 * M * N)
            del X, W, O_, dO, dX, dW

    if "gemmfwd" in only:         # forward only, the two big shapes (tile-order / priority experiments)
        for (M, N, K) in [(4096, 4096, 4096), (8192, 4096, 4096), (16384, 2048, 512)]:
            X, W, b, O_ = rnd(M, K), rnd(N, K) / 64, rnd(N), torch.empty(M, N, device=dev)
            report(f"linear fwd {M}x{K}->{N}", *bench(lambda: call("nnhipLinearModuleForward", X, W, b, O_, M, K, N, st), args.iters), flops=2.0 * M * N * K)
            if M * N * K >= 2 ** 36:
                dO, dX, dW = rnd(M, N), torch.empty(M, K, device=dev), torch.empty(N, K, device=dev)
                report(f"linear dX  {M}x{K}->{N}", *bench(lambda: call("nnhipLinearModuleBackward", X, W, dO, dX, None, None, M, K, N, st), args.iters), flops=2.0 * M * N * K)
                report(f"linear dW  {M}x{K}->{N}", *bench(lambda: call("nnhipLinearModuleBackward", X, W, dO, None, dW, None, M, K, N, st), args.iters), flops=2.0 * M * N * K)
                del dO, dX, dW
            del X, W, O_

    if "headpad" in only:         # does the vocabulary head pay for rows of 15000 floats (60000 B: no multiple of a 128-B line)?
        for (M, N, K) in [(16384, 15000, 512), (16384, 15008, 512), (16384, 15104, 512)]:
            X, W, b = rnd(M, K), rnd(N, K) / 64, rnd(N)
            O_, dO, dX, dW, db = torch.empty(M, N, device=dev), rnd(M, N), torch.empty(M, K, device=dev), torch.empty(N, K, device=dev), torch.empty(N, device=dev)
            fl = 2.0 * M * N * K
            report(f"linear fwd {M}x{K}->{N}", *bench(lambda: call("nnhipLinearModuleForward", X, W, b, O_, M, K, N, st), args.iters), flops=fl)
            report(f"linear dX  {M}x{K}->{N}", *bench(lambda: call("nnhipLinearModuleBackward", X, W, dO, dX, None, None, M, K, N, st), args.iters), flops=fl)
            report(f"linear dW+db {M}x{K}->{N}", *bench(lambda: call("nnhipLinearModuleBackward", X, W, dO, None, dW, db, M, K, N, st), args.iters), flops=fl)
            del X, W, O_, dO, dX, dW

    if "dwsweep" in only:         # the parameter-gradient GEMMs of the C4 step (split-K candidates)
        for (M, N, K) in [(16384, 512, 512), (16384, 1536, 512), (16384, 2048, 512), (16384, 512, 2048), (16384, 15000, 512), (16384, 6144, 512)]:
            X, W, dO = rnd(M, K), rnd(N, K) / 64, rnd(M, N)
            dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
            report(f"linear dW+db {M}x{K}->{N}", *bench(lambda: call("nnhipLinearModuleBackward", X, W, dO, None, dW, db, M, K, N, st), args.iters), flops=2.0 * M * N * K)
            del X, W, dO, dW

    if want("wgroup"):            # a GPT-tiny decoder layer's four dW+db GEMMs: four launches vs the deferred queue's one grid
        M = 16384
        jobs = []
        for (N, K) in [(512, 2048), (2048, 512), (512, 512), (1536, 512)]:      # out, in -- the order backward meets them
            jobs.append((rnd(M, K), rnd(N, K) / 64, rnd(M, N), torch.empty(N, K, device=dev), torch.empty(N, device=dev), K, N))
        fl = sum(2.0 * M * N * K for (*_, K, N) in jobs)

        def separate():
            for (X, W, dO, dW, db, K, N) in jobs:
                call("nnhipLinearModuleBackward", X, W, dO, None, dW, db, M, K, N, st)

        def grouped():
            call("nnhipWeightGradDefer", 1, st)
            for (X, W, dO, dW, db, K, N) in jobs:
                call("nnhipLinearModuleBackward", X, W, dO, None, dW, db, M, K, N, st)
            call("nnhipWeightGradDefer", 0, st)

        report("4 x linear dW+db, GPT-tiny layer (4 launches + 4 reduces)", *bench(separate, args.iters), flops=fl)
        report("4 x linear dW+db, GPT-tiny layer (deferred: 1 grid + 1 reduce)", *bench(grouped, args.iters), flops=fl)
        del jobs

    R, D = 8192, 4096
    n = R * D
    if want("lswish"):
        X, W, b = rnd(R, D), rnd(D, D) / 64, rnd(D)
        O_, Z = torch.empty(R, D, device=dev), torch.empty(R, D, device=dev)
        report("linear_swish fwd (save z)", *bench(lambda: call("nnhipLinearSwishForward", X, W, b, O_, Z, R, D, D, 1.0, 1, st), args.iters), flops=2.0 * R * D * D)
        X5, W5, b5 = rnd(16384, 512), rnd(2048, 512) / 16, rnd(2048)
        O5, Z5 = torch.empty(16384, 2048, device=dev), torch.empty(16384, 2048, device=dev)
        report("linear_swish fwd 16384x512->2048 (save z)", *bench(lambda: call("nnhipLinearSwishForward", X5, W5, b5, O5, Z5, 16384, 512, 2048, 1.0, 1, st), args.iters), flops=2.0 * 16384 * 512 * 2048)
        report("linear fwd 16384x512->2048", *bench(lambda: call("nnhipLinearModuleForward", X5, W5, b5, O5, 16384, 512, 2048, st), args.iters), flops=2.0 * 16384 * 512 * 2048)
        dO5, W6 = rnd(16384, 512), rnd(512, 2048) / 16
        report("dZ = (dO W) swish'(z) 16384x512->2048 (in place)", *bench(lambda: call("nnhipLinearInputGradSwish", dO5, W6, Z5, Z5, 16384, 2048, 512, 1.0, st), args.iters), flops=2.0 * 16384 * 512 * 2048)
        dX5 = torch.empty(16384, 2048, device=dev)
        report("linear dX 16384x2048<-512 (plain)", *bench(lambda: call("nnhipLinearModuleBackward", Z5, W6, dO5, dX5, None, None, 16384, 2048, 512, st), args.iters), flops=2.0 * 16384 * 512 * 2048)
        report("linear_swish fwd (no z)", *bench(lambda: call("nnhipLinearSwishForward", X, W, b, O_, None, R, D, D, 1.0, 0, st), args.iters), flops=2.0 * R * D * D)
        del X, W, O_, Z
    x, dy, y, dx = randn(R, D), randn(R, D), torch.empty(R, D, device=dev), torch.empty(R, D, device=dev)
    if want("swish"):
        report("swish fwd 8192x4096", *bench(lambda: call("nnhipSwishForward", y, x, 1.0, n, st), args.iters), nbytes=8.0 * n)
        report("swish bwd 8192x4096", *bench(lambda: call("nnhipSwishBackward", dx, dy, x, 1.0, n, st), args.iters), nbytes=12.0 * n)
        report("relu fwd 8192x4096", *bench(lambda: call("nnhipReLUForward", y, x, n, st), args.iters), nbytes=8.0 * n)
        report("add 8192x4096", *bench(lambda: call("nnhipAdd", y, x, dy, n, st), args.iters), nbytes=12.0 * n)
    if want("swiglu"):
        o2, d2 = torch.empty(R, D // 2, device=dev), randn(R, D // 2)
        report("swiglu fwd 8192x(2x2048)", *bench(lambda: call("nnhipFusedSwishAndMul", o2, x, 1.0, D // 2, n // 2, st), args.iters), nbytes=12.0 * n / 2)
        report("swiglu bwd 8192x(2x2048)", *bench(lambda: call("nnhipFusedSwishAndMulBackward", dx, d2, x, 1.0, D // 2, n // 2, st), args.iters), nbytes=20.0 * n / 2)
    if want("softmax"):
        report("softmax fwd 8192x4096", *bench(lambda: call("nnhipSoftmaxForward", y, x, R, D, 1, st), args.iters), nbytes=8.0 * n)
        report("softmax bwd 8192x4096", *bench(lambda: call("nnhipSoftmaxBackward", dx, dy, y, R, D, 1, st), args.iters), nbytes=12.0 * n)
        xs = randn(64 * 8 * 256, 256)
        ys = torch.empty_like(xs)
        report("softmax fwd (64,8,256,256)", *bench(lambda: call("nnhipSoftmaxForward", ys, xs, xs.shape[0], 256, 1, st), args.iters), nbytes=8.0 * xs.numel())
    if want("rmsnorm"):
        w, std = torch.ones(D, device=dev), torch.empty(R, device=dev)
        dw = torch.empty(D, device=dev)
        report("rmsnorm fwd 8192x4096", *bench(lambda: call("nnhipRMSNormForward", x, w, None, y, std, None, R, D, 1e-6, st), args.iters), nbytes=8.0 * n)
        report("rmsnorm bwd 8192x4096", *bench(lambda: call("nnhipRMSNormBackward", dy, x, w, std, None, dx, dw, None, R, D, st), args.iters), nbytes=12.0 * n)
        x5, y5, dy5, dx5 = randn(16384, 512), torch.empty(16384, 512, device=dev), randn(16384, 512), torch.empty(16384, 512, device=dev)
        w5, s5, dw5 = torch.ones(512, device=dev), torch.empty(16384, device=dev), torch.empty(512, device=dev)
        report("rmsnorm fwd 16384x512", *bench(lambda: call("nnhipRMSNormForward", x5, w5, None, y5, s5, None, 16384, 512, 1e-6, st), args.iters), nbytes=8.0 * x5.numel())
        report("rmsnorm bwd 16384x512", *bench(lambda: call("nnhipRMSNormBackward", dy5, x5, w5, s5, None, dx5, dw5, None, 16384, 512, st), args.iters), nbytes=12.0 * x5.numel())
        add5 = randn(16384, 512)
        # the form the C4 step launches 12 times: the residual branch's gradient added in the same pass (16 B per element), dw finished later
        report("rmsnorm bwd 16384x512 +addend", *bench(lambda: call("nnhipRMSNormBackwardEx", dy5, x5, w5, s5, None, add5, dx5, dw5, None, 16384, 512, st), args.iters), nbytes=16.0 * x5.numel())
    if want("ce"):
        labels = torch.randint(0, D, (R,), device=dev, dtype=torch.int32)
        loss, lse = torch.empty(R, device=dev), torch.empty(R, device=dev)
        report("ce fwd+bwd 8192x4096", *bench(lambda: call("nnhipCrossEntropyForwardBackward", x, loss, lse, labels, D, -100, R, D, b"m", R, None, dx, st), args.iters), nbytes=8.0 * n)
        lo, cnt = torch.empty((), device=dev), torch.empty(1, device=dev, dtype=torch.int32)
        report("ce loss(mean) 1 launch 8192x4096", *bench(lambda: call("nnhipCrossEntropyLossEx", x, dx, loss, lse, labels, 4, None, D, -100, R, D, b"m", lo, cnt, st), args.iters), nbytes=8.0 * n)
        V = 15000
        xl, dxl = randn(16384, V), torch.empty(16384, V, device=dev)
        lab = torch.randint(0, V, (16384,), device=dev, dtype=torch.int32)
        l2, s2 = torch.empty(16384, device=dev), torch.empty(16384, device=dev)
        report("ce fwd+bwd 16384x15000", *bench(lambda: call("nnhipCrossEntropyForwardBackward", xl, l2, s2, lab, V, 0, 16384, V, b"m", 16384, None, dxl, st), args.iters), nbytes=8.0 * xl.numel())
        report("ce loss(mean) 1 launch 16384x15000", *bench(lambda: call("nnhipCrossEntropyLossEx", xl, dxl, l2, s2, lab, 4, None, V, 0, 16384, V, b"m", lo, cnt, st), args.iters), nbytes=8.0 * xl.numel())
        del xl, dxl
    if want("adamw"):
        p, m, v = randn(R, D), torch.zeros(R, D, device=dev), torch.zeros(R, D, device=dev)
        report("adamw single 8192x4096", *bench(lambda: call("nnhipFusedAdamWStep", p, dy, m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 3, n, 0, 1.0, st), args.iters), nbytes=28.0 * n)
        opt = _lib.load_hip_function("nnhipCreateFusedOptimizer")()
        nt = 200
        ps = [randn(512, 1024) for _ in range(nt)]
        gs = [randn(512, 1024) for _ in range(nt)]
        ms = [torch.zeros(512, 1024, device=dev) for _ in range(nt)]
        vs = [torch.zeros(512, 1024, device=dev) for _ in range(nt)]
        arr = lambda ts: (ctypes.c_void_p * nt)(*[t.data_ptr() for t in ts])  # noqa: E731
        cp, cg, cm, cv = arr(ps), arr(gs), arr(ms), arr(vs)
        cs = (ctypes.c_int64 * nt)(*[t.numel() for t in ps])
        cast = lambda a, t: ctypes.cast(a, ctypes.POINTER(t))  # noqa: E731
        report("adamw multi 200x(512,1024)", *bench(lambda: call(
            "nnhipFusedAdamWMultiTensorStep", opt, nt, cast(cp, ctypes.c_void_p), cast(cg, ctypes.c_void_p),
            cast(cm, ctypes.c_void_p), cast(cv, ctypes.c_void_p), cast(cs, ctypes.c_int64), 1e-3, 0.9, 0.999, 1e-8, 1e-2,
            3, 0, 1.0, st), args.iters), nbytes=28.0 * nt * 512 * 1024)
    if want("attn"):
        # fused attention at the C4 shape (B64 T256 H8 dh64, pad+causal): Q/K/V as column blocks of one [B,T,3D] buffer
        for (B_, T_, H_, dh) in [(64, 256, 8, 64), (16, 1024, 8, 64), (64, 256, 4, 128)]:
            Dm = H_ * dh
            qkv = randn(B_, T_, 3 * Dm)
            dqkv = torch.empty_like(qkv)
            kvalid = torch.ones(B_, T_, dtype=torch.int32, device=dev)
            ctx, dctx = torch.empty(B_, T_, Dm, device=dev), randn(B_, T_, Dm)
            lse = torch.empty(B_, H_, T_, 2, device=dev)
            q_, k_, v_ = (qkv[..., i * Dm:(i + 1) * Dm] for i in range(3))
            dq_, dk_, dv_ = (dqkv[..., i * Dm:(i + 1) * Dm] for i in range(3))
            sc = 1.0 / float(np.sqrt(Dm))
            fwd_fl = 4.0 * B_ * H_ * T_ * T_ * dh / 2          # causal: half the score matrix
            f = lambda: call("nnhipAttentionForward", _lib.StridedView(q_), _lib.StridedView(k_), _lib.StridedView(v_), kvalid,  # noqa: E731
                             ctx, lse, B_, H_, T_, T_, dh, 3 * Dm, sc, 1, st)
            bw = lambda: call("nnhipAttentionBackward", _lib.StridedView(q_), _lib.StridedView(k_), _lib.StridedView(v_), kvalid,  # noqa: E731
                              ctx, dctx, lse, _lib.StridedView(dq_), _lib.StridedView(dk_), _lib.StridedView(dv_), B_, H_, T_, T_, dh,
                              3 * Dm, sc, 1, st)
            report(f"attn fwd B{B_} T{T_} H{H_} dh{dh} causal", *bench(f, args.iters), flops=fwd_fl)
            report(f"attn bwd B{B_} T{T_} H{H_} dh{dh} causal", *bench(bw, args.iters), flops=2.5 * fwd_fl)

    if "attnlayout" in only:
        # the same attention work (512 heads of T256 dh64) from three layouts: fused [B,T,3D] (row stride 6 KB, 256-B head
        # segments), separate [B,T,D] tensors (2 KB stride) and head-major [B*H,T,dh] (contiguous): is the strided gather the cost?
        T_, dh = 256, 64
        sc = 1.0 / float(np.sqrt(512))
        fwd_fl = 4.0 * 512 * T_ * T_ * dh / 2
        for name, B_, H_, packed in [("fused qkv [B,T,3D]", 64, 8, True), ("separate [B,T,D]", 64, 8, False), ("head-major [BH,T,dh]", 512, 1, False)]:
            Dm = H_ * dh
            kvalid = torch.ones(B_, T_, dtype=torch.int32, device=dev)
            ctx, dctx = torch.empty(B_, T_, Dm, device=dev), randn(B_, T_, Dm)
            lse = torch.empty(B_, H_, T_, 2, device=dev)
            if packed:
                qkv = randn(B_, T_, 3 * Dm); dqkv = torch.empty_like(qkv)
                q_, k_, v_ = (_lib.StridedView(qkv[..., i * Dm:(i + 1) * Dm]) for i in range(3))
                dq_, dk_, dv_ = (_lib.StridedView(dqkv[..., i * Dm:(i + 1) * Dm]) for i in range(3))
                LQ = 3 * Dm
            else:
                q_, k_, v_ = randn(B_, T_, Dm), randn(B_, T_, Dm), randn(B_, T_, Dm)
                dq_, dk_, dv_ = torch.empty_like(q_), torch.empty_like(q_), torch.empty_like(q_)
                LQ = Dm
            f = lambda: call("nnhipAttentionForward", q_, k_, v_, kvalid, ctx, lse, B_, H_, T_, T_, dh, LQ, sc, 1, st)  # noqa: E731
            bw = lambda: call("nnhipAttentionBackward", q_, k_, v_, kvalid, ctx, dctx, lse, dq_, dk_, dv_, B_, H_, T_, T_, dh, LQ, sc, 1, st)  # noqa: E731
            report(f"attn fwd {name}", *bench(f, args.iters), flops=fwd_fl)
            report(f"attn bwd {name}", *bench(bw, args.iters), flops=2.5 * fwd_fl)

    if want("lssweep"):
        # The reference's own Linear->Swish sweep (scripts/benchmark_linear_swish_cuda.py:127-138) with its methodology
        # (:16-56, :59-118): 50 warm-up + 200 timed iterations between two events, and INSIDE the timed loop a device copy
        # of x, a fresh Tensor, the module call (forward) or call + fresh random upstream gradient + backward.
        import neunet_hip
        import neunet_hip.nn as nn
        print("Linear->Swish sweep, reference methodology (module API, per-iteration device copy + Tensor + launch; "
              "ms per iteration):")
        print(f"{'B x In x Out':>20s} {'fwd fused':>10s} {'fwd 2-op':>10s} {'f+b fused':>10s} {'f+b 2-op':>10s} {'fwd kernel':>11s} {'TFLOP/s':>8s}")

        def timed(fn, warm=50, iters=200):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / iters

        for (Bn, I, Od) in [(32, 256, 512), (64, 256, 512), (128, 256, 512), (256, 512, 1024), (512, 512, 1024), (1024, 512, 1024),
                            (1024, 1024, 2048), (2048, 1024, 2048), (4096, 1024, 4096)]:
            fused = nn.LinearSwish(I, Od, swish_beta=1.0)
            lin, act = nn.Linear(I, Od), nn.Swish(1.0)
            lin.weight.data.copy_(fused.weight.data)
            lin.bias.data.copy_(fused.bias.data)
            xd = rnd(Bn, I)

            def f_fused():
                return fused(neunet_hip.Tensor(xd.clone(), device="cuda", requires_grad=False))

            def f_two():
                h = lin(neunet_hip.Tensor(xd.clone(), device="cuda", requires_grad=False))
                _ = h.data                                 # the plain GEMM, then the Swish pass: two launches
                return act(h)

            def fb(mod_call, params):
                X = neunet_hip.Tensor(xd.clone(), device="cuda", requires_grad=True)
                out = mod_call(X)
                out.backward(rnd(Bn, Od))
                for p_ in params:
                    p_.grad = None

            def two(X):
                h = lin(X)
                _ = h.data
                return act(h)

            t_ff, t_f2 = timed(f_fused), timed(f_two)
            t_bf = timed(lambda: fb(fused, (fused.weight, fused.bias)))
            t_b2 = timed(lambda: fb(two, (lin.weight, lin.bias)))
            O_ = torch.empty(Bn, Od, device=dev)
            tk, _ = bench(lambda: call("nnhipLinearSwishForward", xd, fused.weight.data, fused.bias.data, O_, None, Bn, I, Od, 1.0, 0, st), 50)
            print(f"{f'{Bn} x {I} x {Od}':>20s} {t_ff:10.4f} {t_f2:10.4f} {t_bf:10.4f} {t_b2:10.4f} {tk:11.4f} {2.0 * Bn * I * Od / (tk * 1e-3) / 1e12:8.2f}", flush=True)

    if want("convg"):
        # the implicit-GEMM conv kernels at MFMA-bound shapes (channels > 16: conv_mfma.hip)
        for (B, Cin, H, Cout) in [(64, 64, 56, 128), (128, 128, 28, 128), (64, 32, 56, 64), (64, 256, 14, 256), (32, 512, 7, 512)]:
            X = rnd(B, Cin, H, H)
            W = rnd(Cout, Cin, 3, 3) / 24
            bb = rnd(Cout)
            O_ = torch.empty(B, Cout, H, H, device=dev)
            dO = rnd(B, Cout, H, H)
            dX = torch.empty_like(X)
            d = Conv2dDesc(B, Cin, H, H, Cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1)
            fl = 2.0 * B * H * H * Cout * Cin * 9
            report(f"conv igemm fwd   {B}x{Cin}x{H}x{H}->{Cout}", *bench(lambda: call("nnhipConv2dForward", X, W, bb, O_, ctypes.byref(d), st), args.iters), flops=fl)
            report(f"conv igemm dgrad {B}x{Cin}x{H}x{H}->{Cout}", *bench(lambda: call("nnhipConv2dBackward", X, W, dO, dX, None, None, ctypes.byref(d), st), args.iters), flops=fl)
            dW, db = torch.empty_like(W), torch.empty_like(bb)
            report(f"conv wgrad+db    {B}x{Cin}x{H}x{H}->{Cout}", *bench(lambda: call("nnhipConv2dBackward", X, W, dO, None, dW, db, ctypes.byref(d), st), args.iters), flops=fl)

    if "convu" in only:           # the reference's DDPM U-Net body (scripts/torch_ddpm.py:451-452: channels 32..512 on 32x32 images), batch 64
        for (B, Cin, H, Cout) in [(64, 32, 32, 64), (64, 64, 16, 128), (64, 128, 8, 256), (64, 256, 4, 512), (64, 512, 2, 512), (64, 512, 4, 256), (64, 128, 16, 64)]:
            X, W, bb = rnd(B, Cin, H, H), rnd(Cout, Cin, 3, 3) / 24, rnd(Cout)
            O_, dO = torch.empty(B, Cout, H, H, device=dev), rnd(B, Cout, H, H)
            dX, dW, db = torch.empty_like(X), torch.empty_like(W), torch.empty_like(bb)
            d = Conv2dDesc(B, Cin, H, H, Cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1)
            fl = 2.0 * B * H * H * Cout * Cin * 9
            report(f"conv fwd      {B}x{Cin}x{H}x{H}->{Cout}", *bench(lambda: call("nnhipConv2dForward", X, W, bb, O_, ctypes.byref(d), st), args.iters), flops=fl)
            report(f"conv dgrad    {B}x{Cin}x{H}x{H}->{Cout}", *bench(lambda: call("nnhipConv2dBackward", X, W, dO, dX, None, None, ctypes.byref(d), st), args.iters), flops=fl)
            report(f"conv wgrad+db {B}x{Cin}x{H}x{H}->{Cout}", *bench(lambda: call("nnhipConv2dBackward", X, W, dO, None, dW, db, ctypes.byref(d), st), args.iters), flops=fl)

    if want("conv"):
        for (B, Cin, H, Cout) in [(256, 1, 28, 8), (256, 8, 14, 16)]:
            X = rnd(B, Cin, H, H)
            W = rnd(Cout, Cin, 3, 3)
            bb = rnd(Cout)
            O_ = torch.empty(B, Cout, H, H, device=dev)
            dO = rnd(B, Cout, H, H)
            dX, dW, db = torch.empty_like(X), torch.empty_like(W), torch.empty_like(bb)
            d = Conv2dDesc(B, Cin, H, H, Cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1)
            by = 4.0 * (X.numel() + O_.numel() + W.numel())
            report(f"conv fwd {B}x{Cin}x{H}x{H}->{Cout}", *bench(lambda: call("nnhipConv2dForward", X, W, bb, O_, ctypes.byref(d), st), args.iters), nbytes=by)
            report(f"conv bwd {B}x{Cin}x{H}x{H}->{Cout}", *bench(lambda: call("nnhipConv2dBackward", X, W, dO, dX, dW, db, ctypes.byref(d), st), args.iters), nbytes=by * 2)


if __name__ == "__main__":
    main()
