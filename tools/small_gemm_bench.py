#!/usr/bin/env python3
"""Kernel time of the small-problem GEMMs of C1 / C5 (GPU box):  python tools/small_gemm_bench.py
Linear forward / input gradient / weight gradient of 32x784->128, 32x128->10 (MNIST-MLP) and 256x784->10 (conv classifier head),
each launched 200 times back to back between two events (NEUNET_HIP_LIB selects the library build)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))

import torch  # noqa: E402

from neunet_hip import _lib  # noqa: E402
from neunet_hip._lib import call_hip_function as call  # noqa: E402


def timed(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


def main():
    st = _lib.get_current_stream_ptr()
    dev = "cuda"
    for (M, K, N) in [(32, 784, 128), (32, 128, 10), (256, 784, 10)]:
        X, W, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
        O, dO = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev)
        dX, dW, db = torch.empty(M, K, device=dev), torch.empty(N, K, device=dev), torch.empty(N, device=dev)
        f = timed(lambda: call("nnhipLinearModuleForward", X, W, b, O, M, K, N, st))
        gx = timed(lambda: call("nnhipLinearModuleBackward", X, W, dO, dX, None, None, M, K, N, st))
        gw = timed(lambda: call("nnhipLinearModuleBackward", X, W, dO, None, dW, db, M, K, N, st))
        print(f"{M}x{K}->{N}: forward {f:6.2f} us   dX {gx:6.2f} us   dW+db {gw:6.2f} us   (back-to-back launches)", flush=True)


if __name__ == "__main__":
    main()
