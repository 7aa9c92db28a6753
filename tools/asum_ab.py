#!/usr/bin/env python3
"""What do the bias gradients cost inside the grouped parameter-gradient launch (DESIGN 5.1b `asum`)?  One decoder layer's four dW
GEMMs of the C4 step as ONE deferred group launch, with db (row sums of dO^T inside the GEMM) and without, HIP-event medians after
half a second of warm-up."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
import torch  # noqa: E402
import neunet_hip  # noqa: E402
from neunet_hip._lib import call_hip_function as call, get_current_stream_ptr  # noqa: E402

neunet_hip.load_library()
st = get_current_stream_ptr()
g = torch.Generator(device="cuda").manual_seed(3)
rnd = lambda *sh: torch.rand(*sh, device="cuda", generator=g) * 2 - 1  # noqa: E731
M, D, F = 16384, 512, 2048
jobs = [(rnd(M, K), rnd(N, K) / 16, rnd(M, N), torch.empty(N, K, device="cuda"), torch.empty(1, N, device="cuda"), K, N)
        for (N, K) in ((D, F), (F, D), (D, D), (3 * D, D))]


def grouped(with_db, which=None):
    call("nnhipWeightGradDefer", 1, st)
    for i, (X, W, dO, dW, db, K, N) in enumerate(jobs):
        use = with_db if which is None else (i in which)
        call("nnhipLinearModuleBackward", X, W, dO, None, dW, db if use else None, M, K, N, st)
    call("nnhipWeightGradDefer", 0, st)


def med(fn, iters=40):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.6:
        fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


for rep in range(2):
    print("with db   ", round(med(lambda: grouped(True)), 1), "us")
    print("without db", round(med(lambda: grouped(False)), 1), "us")
    print("db only for q|k|v", round(med(lambda: grouped(False, which=(3,))), 1), "us")
# what a separate column-sum launch costs for the widest dO (16384 x 2048)
