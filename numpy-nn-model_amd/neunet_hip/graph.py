"""hipGraph capture of a whole training step.

A step of this framework is a few hundred short kernel launches driven from Python (tape replay, ctypes):
at the C1 / C4 sizes the GPU idles while the host enqueues.  MI355X-first answer: run the step eagerly a few
times (warm-up grows the workspace, uploads the optimizer plan, fixes the gradient-bucket addresses), then
capture one step into a hipGraph and replay it -- the Python tape then runs only once, at capture.

    step = GraphedTrainStep(forward_backward, optimizer, bucket)   # forward_backward() -> loss Tensor
    for batch in data:
        static_ids.data.copy_(batch_ids)          # refill the static input buffers
        loss = step()                              # replay

Rules a captured region must obey (all hold for the kernels of libneunet_hip.so): no host synchronisation,
no hipMalloc (the library workspace is grow-only, sized during warm-up and LOCKED after capture so that a later
eager call of a larger shape fails loudly instead of freeing memory the graph points into), static tensor
addresses (inputs live in fixed buffers; parameter gradients live in the flat GradBucket; checked on replay), and
every per-step scalar in device memory: the optimizer's step counter, lr, weight_decay and grad_scale
(`optimizer.use_device_step`; assigning `optimizer.lr` between replays works -- the new value is written to the
device before the next replay) and, for a 'sum' loss under data parallelism, the all-reduced target count
(`optimizer.grad_divisor`).

Shapes of a replayed step:
  * one process:            [ forward + backward + optimizer ]                      one graph, one launch per step
  * N ranks, plain bucket:  [ forward + backward ] -> all_reduce(bucket) -> [ optimizer ]
  * N ranks, GradBucket(overlap=True):
        [ piece 0 ] -> async all_reduce(segment 0) | [ piece 1 ] -> async all_reduce(segment 1) | ... -> wait -> [ optimizer ]
    the backward pass is cut into one graph per bucket segment (at the point the segment's last gradient has been
    written), so RCCL exchanges segment k on its own stream while piece k+1 of the backward pass replays.
"""
from __future__ import annotations

from ._lib import call_hip_function


class GraphedTrainStep:
    _live = 0          # captured steps alive in this process: the library workspace stays locked while > 0

    def __init__(self, forward_backward, optimizer, bucket, warmup: int = 3, world: int = 1, pre_optim=None,
                 group=None, check_every: int = 256, unroll: int = 1):
        import torch
        self.fb, self.opt, self.bucket, self.world = forward_backward, optimizer, bucket, world
        self.pre_optim = pre_optim                    # host-side hook between exchange and optimizer (kept for callers)
        # unroll = U > 1 (one process only): U consecutive steps are captured into ONE graph, so a replay costs the host one
        # launch per U steps.  A sub-0.1 ms step (MNIST-MLP: 7 kernels, 44 us) is otherwise at the mercy of the host's graph
        # launch time.  forward_backward is then called as forward_backward(k), k = 0..U-1, and should read its batch from
        # static input slot k (the caller refills the U slots before every replay); __call__ runs U steps.
        self.unroll = max(1, int(unroll)) if (world == 1 and pre_optim is None) else 1
        self.group = group if group is not None else getattr(bucket, "group", None)
        self._torch = torch
        self._calls = 0
        self._check_every = max(1, int(check_every))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                    # warm-up on a side stream, as torch's graph recipe asks
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if hasattr(optimizer, "use_device_step"):
            optimizer.use_device_step(True)
        self.opt.zero_grad()
        self.single = world == 1 and pre_optim is None
        self.pieces = []                              # [(graph, segment index or None)]
        self.g_opt = None
        overlap = bool(getattr(bucket, "overlap", False)) and not self.single
        if overlap:
            self._capture_overlapped(s)
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for k in range(self.unroll):
                    if k:
                        self.opt.zero_grad()          # host bookkeeping only: the next step's kernels overwrite the gradients
                    self.loss = self.fb(k) if self.unroll > 1 else self.fb()
                    self.bucket.collect()
                    if self.single:
                        self._bind_grads()
                        self.opt.step()
            self.pieces.append((g, None))
        if not self.single:
            self.g_opt = torch.cuda.CUDAGraph()
            self._bind_grads()
            with torch.cuda.graph(self.g_opt, pool=self.pieces[0][0].pool()):
                self.opt.step()
        torch.cuda.synchronize()
        # the captured kernels hold these addresses
        self._addr = self._addresses()
        self._locked = True
        GraphedTrainStep._live += 1
        call_hip_function("nnhipWorkspaceLock", 1)

    # ---- capture of the overlapped variant: one graph per bucket segment ------------------------------------------
    def _capture_overlapped(self, side_stream):
        torch = self._torch
        bucket = self.bucket
        cur = torch.cuda.current_stream()
        side_stream.wait_stream(cur)
        pool = torch.cuda.graph_pool_handle()
        state = {"g": None}

        def begin():
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=pool)
            state["g"] = g

        def cut(k):                                   # called by the bucket where it would launch segment k's all-reduce
            state["g"].capture_end()
            self.pieces.append((state["g"], k))
            begin()

        with torch.cuda.stream(side_stream):
            bucket._capture_cb = cut
            try:
                begin()
                self.loss = self.fb()
                bucket._finish_for_capture()          # zero-fills, late copies, remaining segments (each a cut)
                state["g"].capture_end()
                self.pieces.append((state["g"], None))
            finally:
                bucket._capture_cb = None
        cur.wait_stream(side_stream)

    def release(self):
        """Drop the captured graphs and unlock the library workspace (call when this step object is retired and other
        shapes are going to run)."""
        self.pieces, self.g_opt = [], None
        if getattr(self, "_locked", False):
            self._locked = False
            GraphedTrainStep._live -= 1
            if GraphedTrainStep._live <= 0:
                GraphedTrainStep._live = 0
                call_hip_function("nnhipWorkspaceLock", 0)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def _addresses(self):
        ptrs = [p.data.data_ptr() for p in self.bucket.params]
        ptrs += [v.data_ptr() for v in self.bucket.views]
        return ptrs

    def _bind_grads(self):
        for p, v, hg in zip(self.bucket.params, self.bucket.views, self.bucket.has_grad):
            p.grad = v if hg else None

    def _eager(self):
        self.opt.zero_grad()
        loss = self.fb(0) if self.unroll > 1 else self.fb()
        self.bucket.all_reduce(self.group)
        if self.pre_optim is not None:
            self.pre_optim()
        self.opt.step()
        return loss

    def __call__(self):
        self._calls += 1
        if self._calls <= 3 or self._calls % self._check_every == 0:
            if self._addresses() != self._addr:
                raise RuntimeError("GraphedTrainStep: a parameter or gradient buffer moved since capture (p.data was "
                                   "re-assigned, or the bucket was rebuilt); the captured kernels would read stale "
                                   "memory -- capture again")
        if hasattr(self.opt, "sync_device_hyper"):
            self.opt.sync_device_hyper()              # lr / weight_decay / grad_scale changed on the host -> device state
        if self.single:
            self.pieces[0][0].replay()
            return self.loss
        import torch.distributed as dist
        live = self.world > 1 and dist.is_available() and dist.is_initialized()
        works = []
        for g, k in self.pieces:
            g.replay()
            if k is not None and live:
                lo, hi, _ = self.bucket.segments[k]
                works.append(dist.all_reduce(self.bucket.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if live:
            if getattr(self.bucket, "overlap", False):
                if self.bucket.extra is not None:
                    works.append(dist.all_reduce(self.bucket.flat[self.bucket.extra_offset:], op=dist.ReduceOp.SUM,
                                                 group=self.group, async_op=True))
            else:
                dist.all_reduce(self.bucket.flat, op=dist.ReduceOp.SUM, group=self.group)
        for w in works:
            w.wait()                                  # stream-ordered: the host does not block on the GPU
        if self.pre_optim is not None:
            self.pre_optim()
        self.g_opt.replay()
        return self.loss
