#!/usr/bin/env python3
"""Digests of everything the balanced T = 256 attention kernels (csrc/attention_sb.hip) write, on seeded inputs: the C4 shape (B64 H8,
fused [B,T,3D] layout), an odd number of (batch, head) slices, and the padding patterns the reference's where(mask == 0, -1e9) semantics
distinguishes.  One JSON object on stdout.  Run once per library build (NEUNET_HIP_LIB selects the .so) and compare: the test
test_attention_sb_counted_waits_match_full_drain does that for the default build and the -DSB_CHECK build (every hand-counted
`s_waitcnt vmcnt(N)` replaced by vmcnt(0))."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
import torch  # noqa: E402
from neunet_hip import _lib  # noqa: E402
from neunet_hip.nn.experimental import attention as A  # noqa: E402

os.environ["NNHIP_ATTN_SB"] = "1"
T = 256


def digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:20]


out = {"lib": os.path.basename(_lib.lib_path())}
for B, H, ld3 in ((64, 8, True), (3, 3, True), (2, 5, False), (1, 1, True)):
    D = H * 64
    rng = np.random.default_rng(1000 * B + H)
    pats = {"none": None}
    kv = np.ones((B, T), np.int32); kv[0, -T // 5:] = 0; pats["trailing"] = kv
    pats["holes"] = (rng.random((B, T)) > 0.2).astype(np.int32)
    kv = (rng.random((B, T)) > 0.3).astype(np.int32); kv[0, :140] = 0; pats["leading+holes"] = kv
    for name, kvh in pats.items():
        kvd = None if kvh is None else torch.from_numpy(kvh).cuda()
        if ld3:
            buf = torch.from_numpy(rng.standard_normal((B, T, 3 * D)).astype(np.float32) * 1.5).cuda()
            q, k, v = buf[..., 0:D], buf[..., D:2 * D], buf[..., 2 * D:]
            gb = torch.zeros((B, T, 3 * D), device="cuda")
            outg = (gb[..., 0:D], gb[..., D:2 * D], gb[..., 2 * D:])
        else:
            q, k, v = [torch.from_numpy(rng.standard_normal((B, T, D)).astype(np.float32) * 1.5).cuda() for _ in range(3)]
            outg = None
        do = torch.from_numpy(rng.standard_normal((B, T, D)).astype(np.float32)).cuda()
        scale = float(np.sqrt(D))
        reps = 3 if B == 64 else 1             # the big case a few times: a race does not have to show on the first launch
        for r in range(reps):
            ctx, lse = A.fused_attention_forward(q, k, v, kvd, H, scale, True)
            g = A.fused_attention_backward(q, k, v, kvd, ctx, lse, H, scale, True, do, out=outg)
            torch.cuda.synchronize()
            out[f"B{B} H{H} ld3={int(ld3)} {name} #{r}"] = [digest(ctx), digest(lse)] + [digest(t) for t in g]
print(json.dumps(out))
