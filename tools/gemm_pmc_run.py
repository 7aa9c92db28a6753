#!/usr/bin/env python3
"""Workload for tools/gemm_pmc.sh / gemm_fetch.sh: a few launches of one Linear GEMM (mode from argv[1], shape argv[2:5];
GEMM_PMC_OP = fwd (default) | dx | dw)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
import torch  # noqa: E402
from neunet_hip import _lib  # noqa: E402
from neunet_hip._lib import call_hip_function as call  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
M, K, N = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (8192, 4096, 4096)
call("nnhipSetGemmMode", mode)
st = _lib.get_current_stream_ptr()
X, W = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / 64
O = torch.empty(M, N, device="cuda")
op = os.environ.get("GEMM_PMC_OP", "fwd")
dO = torch.randn(M, N, device="cuda")
dX, dW = torch.empty(M, K, device="cuda"), torch.empty(N, K, device="cuda")
for _ in range(6):
    if op == "fwd":
        call("nnhipLinearModuleForward", X, W, None, O, M, K, N, st)
    elif op == "dx":
        call("nnhipLinearModuleBackward", X, W, dO, dX, None, None, M, K, N, st)
    else:
        call("nnhipLinearModuleBackward", X, W, dO, None, dW, None, M, K, N, st)
torch.cuda.synchronize()
