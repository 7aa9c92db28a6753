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
  * N ranks, capture_collectives=True:
        [ forward + backward + the segments' all-reduces on RCCL's stream (fork/join inside the capture) + optimizer ]
    one graph again: torch's ProcessGroupNCCL records its collectives into an ongoing stream capture, so the exchange
    becomes graph nodes on a parallel branch.  If the capture (or the first replay) fails, the constructor falls back
    to the cut-into-pieces form above; `self.mode` says which one runs ("single" / "pieces" / "ingraph").
"""
from __future__ import annotations

from ._lib import call_hip_function
from .autograd import bump_param_epoch
from .distributed import collectives_live


def attach_step_seed(model):
    """Give every dropout site of `model` (HIPDropout, the fused attention's in-kernel dropout) one shared device uint32 that
    the kernels add to their seed; GraphedTrainStep(step_seed=...) advances it before every replay, so a captured step draws
    fresh masks each time it runs.  Returns the tensor (int32 storage, read as uint32 by the kernels)."""
    import torch
    seed = torch.zeros(1, dtype=torch.int32, device="cuda")
    seen, stack = set(), [model]
    while stack:
        m = stack.pop()
        if id(m) in seen:
            continue
        seen.add(id(m))
        if hasattr(m, "dropout_seed_dev"):
            m.dropout_seed_dev = seed
        if type(m).__name__ == "HIPDropout":
            m.seed_dev = seed
        for v in list(getattr(m, "__dict__", {}).values()):
            if hasattr(v, "__dict__") and not isinstance(v, type) and hasattr(v, "forward"):
                stack.append(v)
            elif isinstance(v, (list, tuple)):
                stack.extend(x for x in v if hasattr(x, "forward"))
            elif hasattr(v, "modules") and isinstance(getattr(v, "modules"), (list, tuple)):
                stack.extend(v.modules)
    return seed


def count_graph_nodes(raw_graph: int):
    """(kernel nodes, all nodes) of a captured hipGraph_t (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()) --
    hipGraphGetNodes + hipGraphNodeGetType of the HIP runtime already mapped into the process."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_size_t(0)
    g = ctypes.c_void_p(raw_graph)
    if hip.hipGraphGetNodes(g, None, ctypes.byref(n)) != 0:
        return None, None
    nodes = (ctypes.c_void_p * max(1, n.value))()
    if hip.hipGraphGetNodes(g, nodes, ctypes.byref(n)) != 0:
        return None, None
    kernels = 0
    for i in range(n.value):
        t = ctypes.c_int(-1)
        if hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) == 0 and t.value == 0:   # hipGraphNodeTypeKernel
            kernels += 1
    return kernels, int(n.value)


def _end_broken_capture(stream):
    """hipStreamEndCapture on a stream whose capture was invalidated: returns the stream to the non-capturing state (the call
    itself reports the invalidation -- that is expected -- and hands back no graph)."""
    import ctypes
    try:
        rt = ctypes.CDLL("libamdhip64.so")
        status = ctypes.c_int(0)
        rt.hipStreamIsCapturing(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(status))
        if status.value != 0:
            g = ctypes.c_void_p()
            rt.hipStreamEndCapture(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(g))
            if g.value:
                rt.hipGraphDestroy(g)
        rt.hipGetLastError()
    except Exception:  # noqa: BLE001 -- best effort: the fall-back uses a fresh stream either way
        pass


class GraphedTrainStep:
    _live = 0          # captured steps alive in this process: the library workspace stays locked while > 0
    # Every capture is thread-local.  torch's default ("global") makes ANY thread's capture-unsafe HIP call fail while this
    # thread captures -- and ProcessGroupNCCL's watchdog thread polls hipEventQuery on the warm-up steps' all-reduce works
    # all the time: the first run of the cut-into-pieces step on a real nccl group died with
    # hipErrorStreamCaptureUnsupported in the watchdog (round 3; gloo has no such thread, which is why it went unseen).
    _CAPTURE_MODE = "thread_local"

    def __init__(self, forward_backward, optimizer, bucket, warmup: int = 3, world: int = 1, pre_optim=None,
                 group=None, check_every: int = 256, unroll: int = 1, capture_collectives: bool = False, step_seed=None,
                 count_nodes: bool = False):
        import torch
        # count_nodes (one-graph steps): keep the hipGraph_t after capture and count its kernel nodes -> self.kernel_nodes
        # (per captured graph, i.e. `unroll` steps) -- what bench.py reports as launches per step
        self._count_nodes, self.kernel_nodes, self.graph_nodes = bool(count_nodes), None, None
        self.fb, self.opt, self.bucket, self.world = forward_backward, optimizer, bucket, world
        self.pre_optim = pre_optim                    # host-side hook between exchange and optimizer (kept for callers)
        # unroll = U > 1 (one process only): U consecutive steps are captured into ONE graph, so a replay costs the host one
        # launch per U steps.  A sub-0.1 ms step (MNIST-MLP: 7 kernels, 44 us) is otherwise at the mercy of the host's graph
        # launch time.  forward_backward is then called as forward_backward(k), k = 0..U-1, and should read its batch from
        # static input slot k (the caller refills the U slots before every replay); __call__ runs U steps.
        self.group = group if group is not None else getattr(bucket, "group", None)
        # "one process" = nothing to exchange: world 1 and no forced collectives (distributed.force_collectives drives the
        # whole DP machinery through a 1-rank group -- the RCCL check on a one-GPU box)
        self.single = world == 1 and pre_optim is None and not collectives_live(self.group)
        self.unroll = max(1, int(unroll)) if self.single else 1
        self._torch = torch
        self.step_seed = step_seed                    # attach_step_seed(model): advanced before every replay
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
        self.pieces = []                              # [(graph, segment index or None)]
        self.g_opt = None
        self.mode = "single" if self.single else "pieces"
        self.ingraph_error = None
        # No garbage collection while this thread captures: a collection that happens to run inside the capture region may
        # finalise an EARLIER step object's hipGraph (its private pool is released with hipFree) -- a capture-unsafe call on the
        # capturing thread that invalidates the capture (seen as hipErrorStreamCaptureInvalidated at the next launch, once in a
        # few runs of the test suite, depending on where the collector's allocation counters stood).
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            self._capture_all(s, bucket, pre_optim, capture_collectives)
        finally:
            if gc_was_on:
                gc.enable()
        torch.cuda.synchronize()
        # the captured kernels hold these addresses
        self._addr = self._addresses()
        self._locked = True
        GraphedTrainStep._live += 1
        call_hip_function("nnhipWorkspaceLock", 1)

    def _capture_all(self, s, bucket, pre_optim, capture_collectives):
        torch = self._torch
        overlap = bool(getattr(bucket, "overlap", False)) and not self.single
        if capture_collectives and not self.single and pre_optim is None and not self._collectives_capturable(bucket):
            # decided up front, not by trying: a capture that a backend invalidates half-way (gloo copies through the host) leaves
            # the thread's capture state poisoned on ROCm 7.2 -- every later HIP call of the process fails with
            # hipErrorStreamCaptureInvalidated, the fall-back included (8-rank rehearsal of bench.py, round 5)
            self.ingraph_error = "the process group's backend cannot record collectives into a stream capture (only nccl = RCCL and the library's NativeComm can)"
            capture_collectives = False
        if capture_collectives and not self.single and pre_optim is None:
            try:
                self._capture_with_collectives(s)
                self.mode = "ingraph"
            except Exception as exc:  # noqa: BLE001 -- the backend refused: keep the collectives between the pieces
                self.ingraph_error = repr(exc)[:300]
                self.pieces = []
                # A refused capture leaves its stream in the INVALIDATED capture state (torch's capture_end raised before the
                # runtime's EndCapture ran): a second capture_begin on it fails half-way and torch then aborts the process from
                # CUDAGraph's destructor ("The graph should be registered to the state") -- found by the 8-rank rehearsal of
                # bench.py, where gloo cannot be captured; on an RCCL node it would have turned "fall back to graph pieces" into a
                # dead job.  End the broken capture by hand and give the fall-back a stream that never saw it.
                _end_broken_capture(s)
                s = torch.cuda.Stream()
                torch.cuda.synchronize()
                if overlap:
                    bucket._reset_step()
                self.opt.zero_grad()
        if self.mode == "ingraph":
            pass
        elif overlap:
            self._capture_overlapped(s)
        else:
            g = torch.cuda.CUDAGraph(keep_graph=True) if self._count_nodes else torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=self._CAPTURE_MODE):
                for k in range(self.unroll):
                    if k:
                        self.opt.zero_grad()          # host bookkeeping only: the next step's kernels overwrite the gradients
                    self.loss = self.fb(k) if self.unroll > 1 else self.fb()
                    if self.single and hasattr(self.opt, "take_pending_backward"):
                        self.opt.take_pending_backward()      # a backward launch that waited for the optimizer: both, now
                    self.bucket.collect()
                    if self.single:
                        self._bind_grads()
                        self.opt.step()
                    if getattr(self.bucket, "overlap", False) and self.bucket.segments:
                        self.bucket._reset_step()     # an overlap bucket on the one-process path: per-step hook state
            if self._count_nodes:
                try:
                    self.kernel_nodes, self.graph_nodes = count_graph_nodes(g.raw_cuda_graph())
                except Exception:  # noqa: BLE001 -- a diagnostic, never a reason to lose the step
                    pass
                g.instantiate()
            self.pieces.append((g, None))
        if self.mode == "pieces":
            self.g_opt = torch.cuda.CUDAGraph()
            self._bind_grads()
            with torch.cuda.graph(self.g_opt, pool=self.pieces[0][0].pool(), capture_error_mode=self._CAPTURE_MODE):
                self.opt.step()

    def _collectives_capturable(self, bucket):
        try:
            from .distributed import NativeComm
            if isinstance(self.group, NativeComm):             # the library's own RCCL entry points: plain stream-ordered launches
                return True
            import torch.distributed as dist
            return dist.is_initialized() and dist.get_backend(self.group) == "nccl"
        except Exception:  # noqa: BLE001
            return False

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
            g.capture_begin(pool=pool, capture_error_mode=self._CAPTURE_MODE)
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

    # ---- capture WITH the collectives: the whole DP step is one graph ------------------------------------------------
    def _capture_with_collectives(self, side_stream):
        torch = self._torch
        g = torch.cuda.CUDAGraph()
        # thread_local: the process group's watchdog thread may touch the runtime while this thread captures
        with torch.cuda.graph(g, stream=side_stream, capture_error_mode=self._CAPTURE_MODE):
            self.loss = self.fb()
            self.bucket.all_reduce(self.group)        # overlap: segments were launched by the hooks; this launches the rest + joins
            self.opt.step()
        torch.cuda.synchronize()
        self.pieces.append((g, None))

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
        if self.single and hasattr(self.opt, "take_pending_backward"):
            self.opt.take_pending_backward()
        self.bucket.all_reduce(self.group)
        if self.pre_optim is not None:
            self.pre_optim()
        self.opt.step()
        return loss

    def __call__(self):
        self._calls += 1
        bump_param_epoch()
        if self.step_seed is not None:
            from ._lib import get_current_stream_ptr
            call_hip_function("nnhipIncrementU32", self.step_seed, 1, get_current_stream_ptr())
        if self._calls <= 3 or self._calls % self._check_every == 0:
            if self._addresses() != self._addr:
                raise RuntimeError("GraphedTrainStep: a parameter or gradient buffer moved since capture (p.data was "
                                   "re-assigned, or the bucket was rebuilt); the captured kernels would read stale "
                                   "memory -- capture again")
        if hasattr(self.opt, "sync_device_hyper"):
            self.opt.sync_device_hyper()              # lr / weight_decay / grad_scale changed on the host -> device state
        if self.single or self.mode == "ingraph":
            self.pieces[0][0].replay()
            return self.loss
        live = collectives_live(self.group)
        works = []
        bk = self.bucket
        for g, k in self.pieces:
            g.replay()
            if k is not None and live:
                lo, hi, _ = bk.segments[k]
                works.append(bk.exchange(lo, hi, async_op=True, group=self.group))
        if live:
            if getattr(bk, "overlap", False):
                if bk.extra is not None:
                    works.append(bk.exchange(bk.extra_offset, bk.numel, async_op=True, group=self.group))
            else:
                bk.exchange(0, bk.numel, group=self.group)
        for w in works:
            w.wait()                                  # stream-ordered: the host does not block on the GPU
        if self.pre_optim is not None:
            self.pre_optim()
        self.g_opt.replay()
        return self.loss
