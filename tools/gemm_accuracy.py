#!/usr/bin/env python3
"""Accuracy of the two GEMM modes against float64 (GPU box):   python tools/gemm_accuracy.py
For each shape: C = X W^T through nnhipLinearModuleForward in mode 0 (exact-fp32 MFMA, v_mfma_f32_32x32x2_f32) and
mode 1 (bf16x3: three-way exact bf16 split, six piece products on v_mfma_f32_32x32x16_bf16), compared with the
float64 product on a sample of rows.  Reported: max and rms of |C - C64| / (|X| |W|^T)  (the error relative to the
magnitude of the summed products -- the quantity fp32 rounding bounds by ~K 2^-24), in units of 2^-24."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))

import torch  # noqa: E402

from neunet_hip import _lib  # noqa: E402
from neunet_hip._lib import call_hip_function as call  # noqa: E402


def main():
    st = _lib.get_current_stream_ptr()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    u = 2.0 ** -24
    for dist in ("uniform(-1,1)", "normal", "lognormal*sign"):
        for (M, N, K) in [(4096, 4096, 4096), (16384, 512, 512), (2048, 15000, 512), (1024, 512, 16384)]:
            if dist == "normal":
                X, W = torch.randn(M, K, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g)
            elif dist == "uniform(-1,1)":
                X, W = torch.rand(M, K, device=dev, generator=g) * 2 - 1, torch.rand(N, K, device=dev, generator=g) * 2 - 1
            else:
                X = torch.exp(2 * torch.randn(M, K, device=dev, generator=g)) * torch.sign(torch.randn(M, K, device=dev, generator=g))
                W = torch.exp(2 * torch.randn(N, K, device=dev, generator=g)) * torch.sign(torch.randn(N, K, device=dev, generator=g))
            rows = torch.arange(0, M, max(1, M // 256), device=dev)[:256]
            C64 = X[rows].double() @ W.double().t()
            mag = X[rows].double().abs() @ W.double().abs().t()
            line = f"{dist:15s} {M}x{K}->{N}:"
            for mode in (0, 1):
                call("nnhipSetGemmMode", mode)
                O = torch.empty(M, N, device=dev)
                call("nnhipLinearModuleForward", X, W, None, O, M, K, N, st)
                torch.cuda.synchronize()
                e = (O[rows].double() - C64).abs() / mag
                line += f"   mode {mode}: max {float(e.max()) / u:7.2f}  rms {float(e.pow(2).mean().sqrt()) / u:6.3f}"
            call("nnhipSetGemmMode", 0)
            print(line + "   (x 2^-24 of sum|x||w|)", flush=True)


if __name__ == "__main__":
    main()
