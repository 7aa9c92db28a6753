#!/usr/bin/env python3
"""Which part of the one-launch CrossEntropy costs what: reduction 'n' (rows only), 's' (+ loss ticket), 'm' (+ label count)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from neunet_hip import _lib  # noqa: E402
from neunet_hip._lib import call_hip_function as call  # noqa: E402
from kbench import bench, report  # noqa: E402

st = _lib.get_current_stream_ptr()
for R, D in [(8192, 4096), (16384, 15000)]:
    x = torch.randn(R, D, device="cuda")
    dx = torch.empty_like(x)
    labels = torch.randint(0, D, (R,), device="cuda", dtype=torch.int32)
    loss, lse = torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
    lo, cnt = torch.empty((), device="cuda"), torch.empty(1, device="cuda", dtype=torch.int32)
    for red in (b"n", b"s", b"m"):
        report(f"ce {R}x{D} reduction={red.decode()}", *bench(lambda: call("nnhipCrossEntropyLossEx", x, dx, loss, lse, labels, 4, None, D, -100, R, D, red, lo, cnt, st), 40), nbytes=8.0 * R * D)
    y = torch.empty_like(x)
    report(f"softmax fwd {R}x{D}", *bench(lambda: call("nnhipSoftmaxForward", y, x, R, D, 1, st), 40), nbytes=8.0 * R * D)
