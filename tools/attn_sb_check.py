#!/usr/bin/env python3
"""Developer tool (GPU box): the balanced 8-wave attention kernels of csrc/attention_sb.hip (causal T = 256, head dim 64)
against the tiled kernels of csrc/attention.hip (NNHIP_ATTN_SB=0) and the GEMM + masked-softmax path, then timings of both
at the C4 shape (B64 H8), fused [B,T,3D] layout.   python tools/attn_sb_check.py [--bwd 0|1] [--iters 40]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from neunet_hip.nn.experimental import attention as A  # noqa: E402
from kbench import bench  # noqa: E402


def sb(on):
    os.environ["NNHIP_ATTN_SB"] = "1" if on else "0"


def key_patterns(B, T, rng):
    pats = {"none": None}
    kv = np.ones((B, T), np.int32)
    kv[0, -T // 5:] = 0
    pats["right-pad"] = kv.copy()
    kv = (rng.random((B, T)) > 0.2).astype(np.int32)
    pats["holes"] = kv.copy()
    kv = np.ones((B, T), np.int32)
    kv[0, :70] = 0
    if B > 1:
        kv[1, 3:9] = 0
    pats["lead-pad"] = kv.copy()
    kv = (rng.random((B, T)) > 0.3).astype(np.int32)
    kv[0, :140] = 0
    pats["lead-pad+holes"] = kv.copy()
    return pats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bwd", type=int, default=1)
    ap.add_argument("--iters", type=int, default=40)
    args = ap.parse_args()
    rng = np.random.default_rng(5)
    torch.manual_seed(0)
    T = 256
    worst = 0.0
    for (B, H, fused_layout) in [(1, 1, False), (2, 3, True), (3, 8, True), (5, 2, False)]:
        D = H * 64
        scale = float(np.sqrt(D))
        for name, kvh in key_patterns(B, T, rng).items():
            kv = None if kvh is None else torch.tensor(kvh, device="cuda")
            if fused_layout:
                buf = torch.randn(B, T, 3 * D, device="cuda") * 1.5
                q, k, v = buf[..., 0:D], buf[..., D:2 * D], buf[..., 2 * D:]
            else:
                q, k, v = [torch.randn(B, T, D, device="cuda") * 1.5 for _ in range(3)]
            do = torch.randn(B, T, D, device="cuda")
            ref, attn, _ = A.attention_forward(q.contiguous(), k.contiguous(), v.contiguous(), kv, H, scale, True)
            gref = A.attention_backward(q.contiguous(), k.contiguous(), v.contiguous(), attn, kv, H, scale, True, do)
            sb(True)
            out, lse = A.fused_attention_forward(q, k, v, kv, H, scale, True)
            err = (out - ref).abs().max().item()
            sb(False)
            out0, lse0 = A.fused_attention_forward(q, k, v, kv, H, scale, True)
            err0 = (out0 - ref).abs().max().item()
            # the new forward's (m, log2 sum) pair must drive the OLD backward to the same gradients (lazy maximum: a consistent pair)
            if fused_layout:
                gb = torch.zeros(B, T, 3 * D, device="cuda")
                outg = (gb[..., 0:D], gb[..., D:2 * D], gb[..., 2 * D:])
            else:
                outg = None
            g_old = [t.clone() for t in A.fused_attention_backward(q, k, v, kv, out, lse, H, scale, True, do, out=outg)]
            eg = max((a - b).abs().max().item() / max(b.abs().max().item(), 1e-6) for a, b in zip(g_old, gref))
            line = f"B{B} H{H} {'3D ' if fused_layout else 'sep'} {name:15s} fwd |sb-ref| {err:.2e} (tiled {err0:.2e})  old-bwd(new lse) rel {eg:.2e}"
            if args.bwd:
                sb(True)
                g_new = [t.clone() for t in A.fused_attention_backward(q, k, v, kv, out, lse, H, scale, True, do, out=outg)]
                g_new2 = A.fused_attention_backward(q, k, v, kv, out, lse, H, scale, True, do, out=outg)
                egn = [(a - b).abs().max().item() / max(b.abs().max().item(), 1e-6) for a, b in zip(g_new, gref)]
                det = all(torch.equal(a, b) for a, b in zip(g_new, g_new2))
                line += f"  sb-bwd rel dq {egn[0]:.2e} dk {egn[1]:.2e} dv {egn[2]:.2e} det={det}"
                worst = max(worst, *egn)
                sb(False)
            worst = max(worst, err, eg)
            print(line, flush=True)
    print(f"WORST {worst:.3e}  {'OK' if worst < 2e-4 else 'FAIL'}", flush=True)

    # ---- timings at the C4 shape -------------------------------------------------------------------------------------
    B, H = 64, 8
    D = H * 64
    scale = float(np.sqrt(D))
    buf = torch.randn(B, T, 3 * D, device="cuda")
    q, k, v = buf[..., 0:D], buf[..., D:2 * D], buf[..., 2 * D:]
    gb = torch.empty_like(buf)
    outg = (gb[..., 0:D], gb[..., D:2 * D], gb[..., 2 * D:])
    kv = torch.ones(B, T, dtype=torch.int32, device="cuda")
    do = torch.randn(B, T, D, device="cuda")
    fl = 4.0 * B * H * T * T * 64 / 2
    sb(True)
    ctx, lse = A.fused_attention_forward(q, k, v, kv, H, scale, True)
    for rep in range(3):
        for on in (False, True):
            sb(on)
            med, mn = bench(lambda: A.fused_attention_forward(q, k, v, kv, H, scale, True), args.iters)
            s = f"fwd {'sb   ' if on else 'tiled'}  med {med * 1e3:7.1f} us  min {mn * 1e3:7.1f} us  {fl / med / 1e9:6.1f} TFLOP/s ({fl / med / 1e9 / 157.3:.3f})"
            if args.bwd:
                med, mn = bench(lambda: A.fused_attention_backward(q, k, v, kv, ctx, lse, H, scale, True, do, out=outg), args.iters)
                s += f" | bwd med {med * 1e3:7.1f} us  min {mn * 1e3:7.1f} us  {2.5 * fl / med / 1e9:6.1f} TFLOP/s ({2.5 * fl / med / 1e9 / 157.3:.3f})"
            print(s, flush=True)
    sb(True)


if __name__ == "__main__":
    main()
