#!/usr/bin/env python3
"""A/B of the persistent GEMM's flush schedule (gemm_pst.hip, NNHIP_PST_STAGGER=0/1, read once per process): the C4 shapes that run
on gemm_pst_kernel, timed with HIP events (median), plus a checksum of every output so that two processes can be compared bit for bit.
    NNHIP_PST_STAGGER=0 python tools/pst_ab.py;  NNHIP_PST_STAGGER=1 python tools/pst_ab.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
import torch  # noqa: E402
import neunet_hip  # noqa: E402
from neunet_hip._lib import call_hip_function as call, get_current_stream_ptr  # noqa: E402

neunet_hip.load_library()
st = get_current_stream_ptr()
g = torch.Generator(device="cuda").manual_seed(11)
rnd = lambda *sh: torch.rand(*sh, device="cuda", generator=g) * 2 - 1  # noqa: E731
M = 16384
iters = int(os.environ.get("PST_AB_ITERS", "30"))


def med(fn):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:          # an idle GPU replays the first few hundred ms at low clocks
        fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


def digest(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


out = {"stagger": os.environ.get("NNHIP_PST_STAGGER", "default(1)")}
for name, K, N in (("qkv_fwd 512->1536", 512, 1536), ("fc1_fwd 512->2048 (plain)", 512, 2048), ("head_fwd 512->15000", 512, 15000),
                   ("k1024 1024->2048", 1024, 2048), ("k256 256->4096", 256, 4096)):
    X, W, b = rnd(M, K), rnd(N, K) / 16, rnd(1, N)
    O_ = torch.empty(M, N, device="cuda")
    us = med(lambda: call("nnhipLinearModuleForward", X, W, b, O_, M, K, N, st))
    out[name] = {"us": round(us, 1), "tflops": round(2.0 * M * K * N / us / 1e6, 1), "sha": digest(O_)}
    if N != 15000:
        dO, dX = rnd(M, N), torch.empty(M, K, device="cuda")
# the Swish epilogues: fc1 forward with z saved (EPI 1) and fc2's input gradient times swish'(z) (EPI 2)
K, N = 512, 2048
X, W, b = rnd(M, K), rnd(N, K) / 16, rnd(1, N)
O_, Z = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
us = med(lambda: call("nnhipLinearSwishForward", X, W, b, O_, Z, M, K, N, 1.0, int(os.environ.get("PST_AB_SWISH_MODE", "2")), st))
out["fc1_fwd+swish (EPI 1)"] = {"us": round(us, 1), "tflops": round(2.0 * M * K * N / us / 1e6, 1), "sha": digest(O_, Z)}
dO, W2 = rnd(M, 512), rnd(512, 2048) / 16          # fc2: 2048 -> 512; dA = dO W2 [M, 2048], times swish'(z) in place over z
Z0 = Z.clone()


def epi2():
    Z.copy_(Z0)
    (call("nnhipLinearInputGradScaled", dO, W2, Z, Z, M, 2048, 512, st) if os.environ.get("PST_AB_SWISH_MODE", "2") == "2" else call("nnhipLinearInputGradSwish", dO, W2, Z, Z, M, 2048, 512, 1.0, st))


def epi2_only():
    (call("nnhipLinearInputGradScaled", dO, W2, Z, Z, M, 2048, 512, st) if os.environ.get("PST_AB_SWISH_MODE", "2") == "2" else call("nnhipLinearInputGradSwish", dO, W2, Z, Z, M, 2048, 512, 1.0, st))


epi2()
sha2 = digest(Z)
us = med(epi2_only)       # (z is overwritten in place: the values drift towards 0, the timing does not care)
out["fc2_dx*swish' (EPI 2)"] = {"us": round(us, 1), "tflops": round(2.0 * M * 2048 * 512 / us / 1e6, 1), "sha": sha2}
print(json.dumps(out, indent=1))
