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
no hipMalloc (the library workspace is grow-only and was sized during warm-up), static tensor addresses
(inputs live in fixed buffers; parameter gradients live in the flat GradBucket), and an optimizer step counter
in device memory (`use_device_step`).  With world_size > 1 the gradient all-reduce stays outside the graphs:
[forward+backward graph] -> all_reduce (RCCL) -> [optimizer graph]; with one process the whole step is ONE graph
(one launch per step instead of two: at MNIST-MLP scale the seam between the two graphs was ~8 us of a 100 us step).
"""
from __future__ import annotations


class GraphedTrainStep:
    def __init__(self, forward_backward, optimizer, bucket, warmup: int = 3, world: int = 1, pre_optim=None):
        import torch
        self.fb, self.opt, self.bucket, self.world = forward_backward, optimizer, bucket, world
        self.pre_optim = pre_optim                    # e.g. set optimizer.grad_scale from an all-reduced count
        self._torch = torch
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                    # warm-up on a side stream, as torch's graph recipe asks
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if hasattr(optimizer, "use_device_step"):
            optimizer.use_device_step(True)
        self.g_fb = torch.cuda.CUDAGraph()
        self.opt.zero_grad()
        self.single = world == 1 and pre_optim is None
        with torch.cuda.graph(self.g_fb):
            self.loss = self.fb()
            self.bucket.collect()
            if self.single:
                self._bind_grads()
                self.opt.step()
        self.g_opt = None
        if not self.single:
            self.g_opt = torch.cuda.CUDAGraph()
            self._bind_grads()
            with torch.cuda.graph(self.g_opt):
                self.opt.step()
        torch.cuda.synchronize()

    def _bind_grads(self):
        for p, v, hg in zip(self.bucket.params, self.bucket.views, self.bucket.has_grad):
            p.grad = v if hg else None

    def _eager(self):
        self.opt.zero_grad()
        loss = self.fb()
        self.bucket.all_reduce()
        if self.pre_optim is not None:
            self.pre_optim()
        self.opt.step()
        return loss

    def __call__(self):
        self.g_fb.replay()
        if self.single:
            return self.loss
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.bucket.flat, op=dist.ReduceOp.SUM)
            if self.pre_optim is not None:
                self.pre_optim()
        self.g_opt.replay()
        return self.loss
