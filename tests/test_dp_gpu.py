"""GPU test of the data-parallel path on a ONE-GPU box: two processes share cuda:0 and exchange the
flat gradient bucket over gloo (the production backend is nccl = RCCL; the code path -- bucket slots written in
place by the HIP layers, one all-reduce, grad_scale folded into the fused optimizer -- is the same).
Claim checked: 2 ranks x half batch, SUM all-reduce, scale 1/world  ==  1 process x full batch."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(hip, seed=123):
    import neunet_hip.nn as nn
    np.random.seed(seed)

    class MLP(nn.Module):
        def __init__(self):
            super().__init__()
            self.l1 = nn.Linear(64, 96)
            self.act = nn.Swish()
            self.norm = nn.RMSNorm(96)
            self.l2 = nn.Linear(96, 10)

        def forward(self, x):
            return self.l2(self.norm(self.act(self.l1(x))))

    return MLP()


def _data():
    rng = np.random.default_rng(9)
    return rng.uniform(-1, 1, (3, 32, 64)).astype(np.float32), rng.integers(0, 10, (3, 32)).astype(np.int32)


def _run(rank, world, port, q, overlap=False):
    import neunet_hip as hip
    import neunet_hip.nn as nn
    from neunet_hip.distributed import GradBucket, shard_batch
    from neunet_hip.optim import AdamW
    if world > 1:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    model = _build(hip)
    params = model.parameters()
    bucket = GradBucket(params, overlap=overlap, segment_bytes=1 << 12)   # overlap: several segments, async all-reduce
    opt = AdamW(params, lr=1e-2, weight_decay=1e-2)
    opt.grad_scale = 1.0 / world
    loss_fn = nn.CrossEntropyLoss()
    X, Y = _data()
    lo, hi = shard_batch(32, rank, world)
    for s in range(3):
        opt.zero_grad()
        out = model(hip.Tensor(X[s, lo:hi], device="cuda", requires_grad=overlap))   # overlap: dW before dX in every Linear
        loss_fn(out, hip.Tensor(Y[s, lo:hi], dtype=np.int32, requires_grad=False, device="cuda")).backward()
        bucket.all_reduce()
        opt.step()
    res = [p.numpy().copy() for p in params]
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    if q is not None:
        q.put((rank, res))
    return res


@pytest.mark.parametrize("overlap", [False, True])
def test_dp2_equals_full_batch(overlap):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = _run(0, 1, 0, None)
    for a, b, r in zip(got[0], got[1], ref):
        np.testing.assert_array_equal(a, b)                         # replicas stay bit-identical
        np.testing.assert_allclose(a, r, rtol=1e-4, atol=1e-5)      # and equal the single-process full batch
