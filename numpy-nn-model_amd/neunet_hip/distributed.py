"""Data-parallel gradient exchange: one process per GPU, one flat fp32 bucket, one all-reduce per step.

Net-new (the reference is single-device).  The bucket holds every parameter gradient in
`Module.parameters()` order (neunet/nn/modules.py:23-39).  Layers write their parameter gradients
straight into their bucket slot (`param._grad_slot`, see experimental/linear.py:_grad_out), so there
is no pack copy; after `all_reduce()` each `param.grad` is a view of the reduced bucket and the fused
optimizer folds the 1/world (or 1/global_count) scale into its load (`grad_scale`).

Backend: torch.distributed -- "nccl" is RCCL on ROCm (xGMI); "gloo" for the CPU tests.
"""
from __future__ import annotations

import os

import numpy as np


def init_process_group(backend: str | None = None):
    """Initialise torch.distributed from the torchrun env (RANK / WORLD_SIZE / MASTER_*). Returns (rank, world)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("NNHIP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            # one process per GPU; LOCAL_RANK wraps so a 1-GPU box can still exercise the N>1 code path (gloo)
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif world == 1 and torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    return rank, world


class GradBucket:
    """Flat gradient bucket over `params` (any objects with .data (torch tensor) and .grad)."""

    def __init__(self, params, extra_scalars: int = 0):
        import torch
        self.params = list(params)
        self.sizes = [int(p.data.numel()) for p in self.params]
        # 16-B aligned slots so the float4 kernels stay on their vector path
        self.offsets, off = [], 0
        for s in self.sizes:
            self.offsets.append(off)
            off += (s + 3) // 4 * 4
        self.extra_offset = off
        self.numel = off + ((extra_scalars + 3) // 4 * 4 if extra_scalars else 0)
        dev = self.params[0].data.device if self.params else "cpu"
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.extra = self.flat[self.extra_offset: self.extra_offset + extra_scalars] if extra_scalars else None
        self.views = [self.flat[o: o + s].view(p.data.shape) for o, s, p in zip(self.offsets, self.sizes, self.params)]
        for p, v in zip(self.params, self.views):
            p._grad_slot = v

    def detach(self):
        for p in self.params:
            if hasattr(p, "_grad_slot"):
                del p._grad_slot

    def collect(self):
        """Make every slot hold this step's local gradient: gradients already written in place are left
        alone; any other gradient is copied in; parameters without a gradient contribute zeros (so every
        rank reduces the same layout, e.g. GPT's never-called cross_attn)."""
        self.has_grad = []
        for p, v in zip(self.params, self.views):
            g = p.grad
            self.has_grad.append(g is not None)
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                v.copy_(g.reshape(v.shape))

    def all_reduce(self, group=None):
        """One SUM all-reduce of the whole bucket (RCCL over xGMI picks direct/tree on the full mesh)."""
        import torch.distributed as dist
        self.collect()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        for p, v, hg in zip(self.params, self.views, self.has_grad):
            p.grad = v if hg else None


def shard_batch(n: int, rank: int, world: int):
    """Even split of a global batch along dim 0 (the remainder goes to the first ranks)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
