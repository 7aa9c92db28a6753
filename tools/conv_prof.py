#!/usr/bin/env python3
"""One conv layer forward / dgrad / wgrad in a loop (for rocprofv3 --kernel-trace --stats): tools/conv_prof.py B Cin H Cout [iters]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
import torch  # noqa: E402
from neunet_hip._lib import Conv2dDesc, call_hip_function as call, get_current_stream_ptr, load_library  # noqa: E402

load_library()
B, Cin, H, Cout = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) * 2 - 1  # noqa: E731
X, W, bb = rnd(B, Cin, H, H), rnd(Cout, Cin, 3, 3) / 24, rnd(Cout)
O_, dO = torch.empty(B, Cout, H, H, device="cuda"), rnd(B, Cout, H, H)
dX, dW, db = torch.empty_like(X), torch.empty_like(W), torch.empty_like(bb)
d = Conv2dDesc(B, Cin, H, H, Cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1)
st = get_current_stream_ptr()
for _ in range(iters):
    call("nnhipConv2dForward", X, W, bb, O_, ctypes.byref(d), st)
    call("nnhipConv2dBackward", X, W, dO, dX, None, None, ctypes.byref(d), st)
    call("nnhipConv2dBackward", X, W, dO, None, dW, db, ctypes.byref(d), st)
torch.cuda.synchronize()
