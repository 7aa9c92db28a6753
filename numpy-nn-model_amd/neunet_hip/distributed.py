"""Data-parallel gradient exchange: one process per GPU, one flat fp32 bucket, one all-reduce per step.

Net-new (the reference is single-device).  The bucket holds every parameter gradient in
`Module.parameters()` order (neunet/nn/modules.py:23-39).  Layers write their parameter gradients
straight into their bucket slot (`param._grad_slot`, see experimental/linear.py:_grad_out), so there
is no pack copy; after `all_reduce()` each `param.grad` is a view of the reduced bucket and the fused
optimizer folds the 1/world (or 1/global_count) scale into its load (`grad_scale`).

With `overlap=True` the bucket is cut into segments (in reverse parameter order, i.e. the order gradients
arrive in) and each segment's all-reduce is launched asynchronously the moment its last gradient has been
written: layers compute their parameter gradients BEFORE their input gradient (experimental/linear.py), so
the exchange of dW rides under the dX GEMM and the rest of the backward pass; `all_reduce()` then only
launches what is left and waits.  RCCL runs the collective on its own stream, ordered after the producing
kernels by an event.

Backend: torch.distributed -- "nccl" is RCCL on ROCm (xGMI); "gloo" for the CPU tests.
"""
from __future__ import annotations

import os

# A ONE-rank process group normally skips every collective (nothing to exchange).  With this switch on (NNHIP_FORCE_DP=1,
# `bench.py --force-dp`, tests) the data-parallel machinery runs all the same: bucket segments, graph cuts, asynchronous
# all-reduces on the backend's own stream, 'sum' loss + device-side divisor.  It is how the RCCL path is driven on a
# one-GPU box: `init_process_group("nccl", force=True)` makes a 1-rank RCCL communicator.
force_collectives = os.environ.get("NNHIP_FORCE_DP", "0") == "1"


def collectives_live(group=None) -> bool:
    """True when a gradient exchange has to be issued: an initialised process group with more than one rank, or with
    one rank and `force_collectives` set."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or force_collectives


def init_process_group(backend: str | None = None, force: bool = False):
    """Initialise torch.distributed from the torchrun env (RANK / WORLD_SIZE / MASTER_*). Returns (rank, world).
    force=True: create the group even for one rank and switch `force_collectives` on."""
    import torch
    import torch.distributed as dist
    global force_collectives
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if force:
        force_collectives = True
    if (world > 1 or force) and not dist.is_initialized():
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

    def __init__(self, params, extra_scalars: int = 0, overlap: bool = False, segment_bytes: int = 32 << 20,
                 group=None, reduce_op: str = "sum"):
        import torch
        # "sum" (default; the optimizer folds 1/world or 1/global_count into its gradient load) or "avg" (RCCL pre-scales
        # by 1/world inside the collective: numerator and extra-slot count shrink together, so g/count is unchanged and
        # plain mean losses need no grad_scale).  On a forced 1-rank group "avg" is also what makes RCCL launch a device
        # kernel at all -- a 1-rank in-place SUM is elided inside the library.
        if reduce_op not in ("sum", "avg"):
            raise ValueError(f"reduce_op must be 'sum' or 'avg', got {reduce_op!r}")
        self.reduce_op = reduce_op
        self.params = list(params)
        self.overlap = overlap
        self.group = group
        self.sizes = [int(p.data.numel()) for p in self.params]
        # Slot layout: `Module.parameters()` order, except that a parameter carrying `_bucket_group` (a list of
        # parameters, e.g. the q/k/v projection weights of a fused attention block) pulls its group next to itself so a
        # fused kernel can write all their gradients as one matrix.  Every rank builds the same model -> same layout.
        index = {id(p): i for i, p in enumerate(self.params)}
        self.layout, placed = [], set()
        for i, p in enumerate(self.params):
            if i in placed:
                continue
            group = [index[id(m)] for m in getattr(p, "_bucket_group", ()) if id(m) in index]
            for j in (group if i in group else [i]):
                if j not in placed:
                    placed.add(j)
                    self.layout.append(j)
        # 16-B aligned slots so the float4 kernels stay on their vector path
        self.offsets, off = [0] * len(self.params), 0
        for i in self.layout:
            self.offsets[i] = off
            off += (self.sizes[i] + 3) // 4 * 4
        self.extra_offset = off
        self.numel = off + ((extra_scalars + 3) // 4 * 4 if extra_scalars else 0)
        dev = self.params[0].data.device if self.params else "cpu"
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.extra = self.flat[self.extra_offset: self.extra_offset + extra_scalars] if extra_scalars else None
        self.views = [self.flat[o: o + s].view(p.data.shape) for o, s, p in zip(self.offsets, self.sizes, self.params)]
        for p, v in zip(self.params, self.views):
            p._grad_slot = v
        # ---- overlap: segments of >= segment_bytes, walking the parameters backwards ----------------------
        self.segments = []        # (lo, hi, [param indices])  -- flat[lo:hi]
        self._seg_of = {}
        self._pending, self._works, self._launched = [], [], []
        self._capture_cb = None       # GraphedTrainStep: called instead of launching a segment's all-reduce (graph cut)
        if overlap:
            hi, idxs = self.extra_offset, []
            for n, i in enumerate(reversed(self.layout)):
                idxs.append(i)
                if (hi - self.offsets[i]) * 4 >= segment_bytes or n == len(self.layout) - 1:
                    self.segments.append((self.offsets[i], hi, idxs))
                    hi, idxs = self.offsets[i], []
            for k, (_, _, idxs) in enumerate(self.segments):
                for i in idxs:
                    self._seg_of[id(self.params[i])] = k
            for p in self.params:
                p._grad_hook = self._on_grad
            self._reset_step()

    def _reset_step(self):
        self._pending = [len(idxs) for (_, _, idxs) in self.segments]
        self._launched = [False] * len(self.segments)
        self._seen = set()
        self._works = []

    def _live(self):
        return collectives_live(self.group)

    def op(self):
        import torch.distributed as dist
        return dist.ReduceOp.AVG if self.reduce_op == "avg" else dist.ReduceOp.SUM

    def _launch(self, k):
        import torch.distributed as dist
        from ._lib import wgrad_flush
        wgrad_flush()                         # queued parameter-gradient GEMMs of this segment go out before it is exchanged
        self._launched[k] = True
        if self._capture_cb is not None:      # hipGraph capture: the step is cut here; the replay launches the exchange
            self._capture_cb(k)
            return
        if self._live():
            lo, hi, _ = self.segments[k]
            self._works.append(dist.all_reduce(self.flat[lo:hi], op=self.op(), group=self.group, async_op=True))

    def _on_grad(self, param):
        """Called by a layer right after it wrote `param`'s gradient (into the slot, or elsewhere -> copied in)."""
        if id(param) in self._seen:
            raise RuntimeError("GradBucket(overlap=True): a parameter received a second gradient after its first was "
                               "handed to the all-reduce; use overlap=False for models that share parameters")
        self._seen.add(id(param))
        k = self._seg_of[id(param)]
        v = param._grad_slot
        g = param.grad
        if g is not None and g.data_ptr() != v.data_ptr():
            from ._lib import wgrad_flush
            wgrad_flush()                     # the gradient may still be a queued GEMM
            v.copy_(g.reshape(v.shape))
            param.grad = v
        self._pending[k] -= 1
        if self._pending[k] == 0 and not self._launched[k]:
            self._launch(k)

    def detach(self):
        for p in self.params:
            if hasattr(p, "_grad_slot"):
                del p._grad_slot
            if hasattr(p, "_grad_hook"):
                del p._grad_hook

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

    def _close_segments(self):
        """Make every slot final (parameters without a gradient are zero-filled, gradients written elsewhere are copied
        in) and launch the segments that are still open."""
        self.has_grad = []
        for p, v in zip(self.params, self.views):
            hg = p.grad is not None
            self.has_grad.append(hg)
            if id(p) not in self._seen:
                if not hg:
                    v.zero_()
                elif p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad.reshape(v.shape))
            elif not hg or p.grad.data_ptr() != v.data_ptr():
                raise RuntimeError("GradBucket(overlap=True): a parameter's gradient changed after it was handed to the "
                                   "all-reduce (shared parameter?); use overlap=False")
        for k in range(len(self.segments)):
            if not self._launched[k]:
                self._launch(k)

    def _finish_for_capture(self):
        """End of a captured backward pass (GraphedTrainStep): like _finish_overlapped without any collective."""
        self._close_segments()
        for p, v, hg in zip(self.params, self.views, self.has_grad):
            p.grad = v if hg else None
        self._reset_step()

    def _finish_overlapped(self):
        """Launch the segments that are still open (parameters without a gradient are zero-filled), exchange the
        extra scalars, wait for everything (stream-ordered: the host does not block on the GPU)."""
        import torch.distributed as dist
        self._close_segments()
        if self.extra is not None and self._live():
            self._works.append(dist.all_reduce(self.flat[self.extra_offset:], op=self.op(), group=self.group,
                                               async_op=True))
        for w in self._works:
            w.wait()
        for p, v, hg in zip(self.params, self.views, self.has_grad):
            p.grad = v if hg else None
        self._reset_step()

    def all_reduce(self, group=None):
        """One SUM all-reduce of the whole bucket (RCCL over xGMI picks direct/tree on the full mesh); with
        overlap=True: launch what the backward pass has not launched yet, then wait."""
        import torch.distributed as dist
        if self.overlap:
            return self._finish_overlapped()
        group = group if group is not None else self.group     # a bucket built for a sub-group reduces over THAT group
        self.collect()
        if collectives_live(group):
            dist.all_reduce(self.flat, op=self.op(), group=group)
        for p, v, hg in zip(self.params, self.views, self.has_grad):
            p.grad = v if hg else None


def shard_batch(n: int, rank: int, world: int):
    """Even split of a global batch along dim 0 (the remainder goes to the first ranks)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
