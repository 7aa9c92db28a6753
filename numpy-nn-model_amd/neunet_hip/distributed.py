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

Backends: torch.distributed -- "nccl" is RCCL on ROCm (xGMI); "gloo" for the CPU tests -- or `NativeComm`, an RCCL
communicator held through the library's own C ABI (nnhipComm*, include/neunet_hip.h): `GradBucket(params, comm=NativeComm.
from_env())` exchanges through nnhipAllReduceSumF32 on a side stream of this process, no ProcessGroup involved.
"""
from __future__ import annotations

import os

# A ONE-rank process group normally skips every collective (nothing to exchange).  With this switch on (NNHIP_FORCE_DP=1,
# `bench.py --force-dp`, tests) the data-parallel machinery runs all the same: bucket segments, graph cuts, asynchronous
# all-reduces on the backend's own stream, 'sum' loss + device-side divisor.  It is how the RCCL path is driven on a
# one-GPU box: `init_process_group("nccl", force=True)` makes a 1-rank RCCL communicator.
force_collectives = os.environ.get("NNHIP_FORCE_DP", "0") == "1"


def collectives_live(group=None) -> bool:
    """True when a gradient exchange has to be issued: an initialised process group (or a NativeComm passed as `group`)
    with more than one rank, or with one rank and `force_collectives` set."""
    if isinstance(group, NativeComm):
        return group.world > 1 or force_collectives
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
        _lockstep_for_collectives(world)
    elif world == 1 and torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    return rank, world


def _lockstep_for_collectives(world: int):
    """More than one rank: the GEMMs share the Infinity Fabric with RCCL's xGMI traffic, so the exact-fp32 kernel runs in its
    lock-step mode by default (nnhipSetGemmLockstep: the two blocks of a CU keep step, operand panels are fetched from the fabric
    once instead of twice -- 4096^3 forward 808 -> 575 MB of fabric reads, bit-identical results, +0.6 ... +2.7 % kernel time on
    reductions >= 2048).  NNHIP_GEMM_LOCKSTEP=0/1 in the environment decides instead when set."""
    if world <= 1 or "NNHIP_GEMM_LOCKSTEP" in os.environ:
        return
    try:
        import torch
        if torch.cuda.is_available():
            from ._lib import call_hip_function
            call_hip_function("nnhipSetGemmLockstep", 1)
    except Exception:  # noqa: BLE001 -- a tuning default, never a reason to fail the start-up
        pass


class _StreamWork:
    """Handle of a collective enqueued on a side stream: wait() orders torch's current stream after it (stream-ordered, the
    host does not block) -- the same contract as the Work object dist.all_reduce(async_op=True) returns."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        import torch
        torch.cuda.current_stream().wait_event(self.event)


class _NoWork:
    """Handle of an exchange that was not issued (GradBucket.mute)."""

    def wait(self):
        pass


class NativeComm:
    """An RCCL communicator held through libneunet_hip.so's C ABI (nnhipCommUniqueId / nnhipCommInitRank /
    nnhipAllReduceSumF32 / nnhipBroadcastF32 / nnhipCommDestroy) -- the binding a reference-side caller with plain device
    pointers would use ("data-parallel loop").  torch is only the allocator and the stream here.

        comm = NativeComm.from_env()                 # RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT
        bucket = GradBucket(params, comm=comm)       # or comm.all_reduce(tensor) directly
    """

    def __init__(self, unique_id: bytes, rank: int, world: int):
        import ctypes
        from ._lib import call_hip_function
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes nnhipCommUniqueId wrote on rank 0")
        self.rank, self.world = int(rank), int(world)
        h = ctypes.c_void_p()
        call_hip_function("nnhipCommInitRank", ctypes.byref(h), unique_id, self.rank, self.world)
        self._h = h
        self._stream = None

    @staticmethod
    def unique_id() -> bytes:
        import ctypes
        from ._lib import call_hip_function
        buf = ctypes.create_string_buffer(128)
        call_hip_function("nnhipCommUniqueId", buf)
        return buf.raw

    @staticmethod
    def library():
        """(path, version) of the librccl the C ABI bound."""
        import ctypes
        from ._lib import call_hip_function
        buf, v = ctypes.create_string_buffer(256), ctypes.c_int(0)
        call_hip_function("nnhipCommLibrary", buf, 256, ctypes.byref(v))
        return buf.value.decode(), int(v.value)

    @classmethod
    def from_env(cls, port_offset: int = 17):
        """One process per GPU under torchrun (or a lone process: rank 0 of 1).  Rank 0 draws the id; it travels through
        the torch process group when one is up, else through a TCPStore on MASTER_PORT + port_offset (rendezvous only --
        no torch collective is created)."""
        import torch
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
            torch.cuda.current_stream()                 # the HIP context exists before RCCL asks for the device
        if world == 1:
            return cls(cls.unique_id(), 0, 1)
        _lockstep_for_collectives(world)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            box = [cls.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
        else:
            from datetime import timedelta
            store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")) + port_offset,
                                  world, rank == 0, timeout=timedelta(seconds=120))
            if rank == 0:
                store.set("nnhip_comm_id", cls.unique_id())
            uid = bytes(store.get("nnhip_comm_id"))
        return cls(uid, rank, world)

    # ---- collectives (in place, fp32) --------------------------------------------------------------------------------------
    def _check(self, t):
        import torch
        if self._h is None:
            raise RuntimeError("NativeComm: communicator already destroyed")
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise TypeError("NativeComm collectives take contiguous fp32 device tensors")

    def all_reduce(self, t, op: str = "sum", async_op: bool = False):
        """In-place all-reduce of `t`.  async_op=False: on torch's current stream, in order with the producers.
        async_op=True: on this communicator's side stream, ordered after everything enqueued on the current stream so far;
        returns a handle whose wait() orders the current stream after the collective (overlap with later kernels)."""
        import torch
        from ._lib import call_hip_function
        self._check(t)
        name = {"sum": "nnhipAllReduceSumF32", "avg": "nnhipAllReduceAvgF32"}[op]
        if not async_op:
            call_hip_function(name, self._h, t, t.numel(), torch.cuda.current_stream().cuda_stream)
            return None
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        ready = torch.cuda.Event()
        ready.record()
        self._stream.wait_event(ready)
        call_hip_function(name, self._h, t, t.numel(), self._stream.cuda_stream)
        done = torch.cuda.Event()
        done.record(self._stream)
        return _StreamWork(done)

    def broadcast(self, t, root: int = 0):
        import torch
        from ._lib import call_hip_function
        self._check(t)
        call_hip_function("nnhipBroadcastF32", self._h, t, t.numel(), int(root), torch.cuda.current_stream().cuda_stream)

    def destroy(self):
        if getattr(self, "_h", None) is not None:
            from ._lib import call_hip_function
            import torch
            torch.cuda.synchronize()
            h, self._h = self._h, None
            call_hip_function("nnhipCommDestroy", h)

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class GradBucket:
    """Flat gradient bucket over `params` (any objects with .data (torch tensor) and .grad)."""

    def __init__(self, params, extra_scalars: int = 0, overlap: bool = False, segment_bytes: int = 32 << 20,
                 group=None, reduce_op: str = "sum", comm: "NativeComm | None" = None):
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
        # comm = a NativeComm: the exchange goes through the library's own RCCL entry points (nnhipAllReduce*F32) instead of
        # torch.distributed; it rides in `group` so that every "is a collective live / which ranks" question has one answer
        if comm is not None and group is not None:
            raise ValueError("GradBucket: pass either a torch process group or a NativeComm, not both")
        self.group = comm if comm is not None else group
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
        # mute = True: exchange() issues nothing (measurement only -- bench.py times the same step with and without its
        # collectives to report the EXPOSED all-reduce time; replicas drift apart while it is set)
        self.mute = False
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

    def exchange(self, lo: int, hi: int, async_op: bool = False, group=None):
        """All-reduce flat[lo:hi] over the bucket's backend (a NativeComm or a torch process group); async_op=True returns a
        handle with wait().  The one place a collective is issued: eager steps, hooks and graph replays all come here."""
        group = group if group is not None else self.group
        if self.mute:
            return _NoWork() if async_op else None
        if isinstance(group, NativeComm):
            return group.all_reduce(self.flat[lo:hi], op=self.reduce_op, async_op=async_op)
        import torch.distributed as dist
        if self.reduce_op == "avg" and dist.get_backend(group) == "gloo":
            raise ValueError("GradBucket(reduce_op='avg'): gloo has no AVG reduction; use 'sum' (the optimizer's grad_scale "
                             "= 1/world gives the mean)")
        return dist.all_reduce(self.flat[lo:hi], op=self.op(), group=group, async_op=async_op)

    def exchange_pieces(self):
        """The (lo, hi) ranges one step exchanges: the overlap segments + the extra scalars, or the whole bucket."""
        if self.overlap and self.segments:
            out = [(lo, hi) for (lo, hi, _) in self.segments]
            if self.extra is not None:
                out.append((self.extra_offset, self.numel))
            return out
        return [(0, self.numel)]

    def time_exchange_alone(self, repeats: int = 5):
        """Every all-reduce of one step issued ON ITS OWN (no compute next to it), each bracketed by two events on the current
        stream -- the collective's own stream starts after the first and the current stream joins it before the second, for both
        backends -- `repeats` times; returns [(floats, median ms)] per piece.  Every rank must call it (they are collectives).
        The contents of the bucket are multiplied by the world size each time (SUM): measurement only."""
        import time
        import torch
        res = []
        on_gpu = self.flat.is_cuda
        for lo, hi in self.exchange_pieces():
            spans = []
            for _ in range(max(1, repeats)):
                if on_gpu:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    self.exchange(lo, hi)
                    b.record()
                    spans.append((a, b))
                else:                         # gloo on host tensors (the CPU tests): the call blocks until the result is there
                    t0 = time.perf_counter()
                    self.exchange(lo, hi)
                    spans.append((time.perf_counter() - t0) * 1e3)
            if on_gpu:
                torch.cuda.synchronize()
                ms = sorted(x.elapsed_time(y) for x, y in spans)
            else:
                ms = sorted(spans)
            res.append((hi - lo, ms[len(ms) // 2]))
            self.flat[lo:hi].zero_()          # (SUM x repeats of random gradients: keep the values finite for the next call)
        return res

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
            self._works.append(self.exchange(lo, hi, async_op=True))

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
            self._works.append(self.exchange(self.extra_offset, self.numel, async_op=True))
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
            self.exchange(0, self.numel, group=group)
        for p, v, hg in zip(self.params, self.views, self.has_grad):
            p.grad = v if hg else None


def shard_batch(n: int, rank: int, world: int):
    """Even split of a global batch along dim 0 (the remainder goes to the first ranks)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
