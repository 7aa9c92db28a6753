#!/usr/bin/env python3
"""Developer soak test: the README MLP trained for many graph-replayed steps with the optimizer inside the backward launch
(arrival counters, polling blocks, stepper block; both the round-5 default path and the round-3 opt-in) against the same run with
the separate optimizer launch -- parameters and
optimizer state must stay BIT-IDENTICAL after every chunk of steps.  usage: python tools/soak_c1.py [steps] [chunk]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
import neunet_hip  # noqa: E402,F401
import neunet_hip.nn as nn  # noqa: E402
import torch  # noqa: E402
from neunet_hip import Tensor, optim  # noqa: E402
from neunet_hip.distributed import GradBucket  # noqa: E402
from neunet_hip.graph import GraphedTrainStep  # noqa: E402


class MLP(nn.Module):
    def __init__(self):
        super().__init__()
        self.l1, self.relu, self.l2 = nn.Linear(784, 128), nn.ReLU(), nn.Linear(128, 10)

    def forward(self, x):
        return self.l2(self.relu(self.l1(x)))


def build(mode, X, Y):
    """mode: "default" (round 5: the backward launch waits for optimizer.step() and takes Adam with it), "opt-in"
    (optimizer.fuse_backward(True), round 3) or "two-launches" (NNHIP_AUTO_FUSE_STEP=0: the reference's sequence)."""
    import neunet_hip.nn.experimental.linear as L
    L._AUTO_FUSE_STEP = mode != "two-launches"               # read at capture time (GraphedTrainStep captures in here)
    np.random.seed(11)
    model = MLP()
    ps = model.parameters()
    opt = optim.Adam(ps, lr=1e-3)
    if mode == "opt-in":
        opt.fuse_backward(True)
    x = Tensor(X[0], device="cuda", requires_grad=False)
    y = Tensor(Y[0], dtype=np.int32, device="cuda", requires_grad=False)
    loss_fn = nn.CrossEntropyLoss()

    def fb():
        loss = loss_fn(model(x), y)
        loss.backward()
        return loss

    return GraphedTrainStep(fb, opt, GradBucket(ps), warmup=1), ps, opt, x, y


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    rng = np.random.default_rng(3)
    X = rng.uniform(-1, 1, (64, 32, 784)).astype(np.float32)
    Y = rng.integers(0, 10, (64, 32)).astype(np.int32)
    Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
    b = build("two-launches", X, Y)
    a = build("default", X, Y)
    c = build("opt-in", X, Y)
    done = 0
    while done < steps:
        for (step, ps, opt, x, y) in (a, b, c):
            for s in range(chunk):
                i = (done + s) % 64
                x.data.copy_(Xd[i])
                y.data.copy_(Yd[i])
                step()
        done += chunk
        torch.cuda.synchronize()
        same = all(all(torch.equal(p.data, q.data) for p, q in zip(o[1], b[1])) and
                   all(torch.equal(m1, m2) for m1, m2 in zip(o[2].m, b[2].m)) and
                   all(torch.equal(v1, v2) for v1, v2 in zip(o[2].v, b[2].v)) for o in (a, c))
        print(f"{done:7d} steps: {'bit-identical' if same else 'MISMATCH'}", flush=True)
        if not same:
            sys.exit(1)
    print("soak ok")


if __name__ == "__main__":
    main()
