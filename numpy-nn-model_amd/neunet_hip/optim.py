"""Fused optimizers -- drop-ins for CUDAFusedAdamW (experimental/optim/fused_adamw/fused_adamw.py:45-113)
and CUDAFusedMultiTensorAdamW (.../fused_adamw_multitensor.py:47-148), plus `Adam` / `AdamW` names that
select the L2-on-gradient (neunet/optim.py:17-33) or decoupled (optim.py:52-69) decay mode of the same
fused kernel."""
import ctypes
import weakref
from ctypes import c_int64, c_void_p

from ._lib import call_hip_function, get_current_stream_ptr, load_hip_function
from .autograd import bump_param_epoch

DECOUPLED, L2_ON_GRAD = 0, 1


def _check_params(params):
    import torch
    for p in params:
        if p.device != "cuda":
            raise ValueError("Fused AdamW only supports parameters on the HIP device ('cuda').")
        if p.data.dtype != torch.float32:
            raise ValueError(f"Fused AdamW only supports float32 parameters, got {p.data.dtype}")


class HIPFusedAdamW:
    """One launch per tensor (fused_adamw.py:64-109)."""
    decay_mode = DECOUPLED

    def __init__(self, params, lr: float = 0.01, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01):
        import torch
        self.params = list(params)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        _check_params(self.params)
        self.m = [torch.zeros_like(p.data) for p in self.params]
        self.v = [torch.zeros_like(p.data) for p in self.params]
        self.t = 0
        self.grad_scale = 1.0

    def step(self):
        self.t += 1
        bump_param_epoch()        # parameters change in place: deferred Linear outputs of this step are now stale
        stream = get_current_stream_ptr()
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            if not p.data.is_contiguous():
                p.data = p.data.contiguous()
            call_hip_function("nnhipFusedAdamWStep", p.data, g, self.m[i], self.v[i], self.lr, self.betas[0],
                              self.betas[1], self.eps, self.weight_decay, self.t, p.data.numel(),
                              self.decay_mode, self.grad_scale, stream)

    def zero_grad(self):
        for p in self.params:
            p.grad = None


class HIPFusedMultiTensorAdamW:
    """One launch for all tensors (fused_adamw_multitensor.py:88-144).  Pointer tables are ctypes arrays in
    host memory whose entries are device pointers -- same contract as the reference."""
    decay_mode = DECOUPLED

    def __init__(self, params, lr: float = 0.01, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01):
        import torch
        self.params = list(params)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.t = 0
        self.grad_scale = 1.0
        _check_params(self.params)
        self.m = [torch.zeros_like(p.data) for p in self.params]
        self.v = [torch.zeros_like(p.data) for p in self.params]
        self.device_step = False      # True: step count, lr, weight_decay, grad_scale live on the device (hipGraph replay)
        self._dev_hyper = None        # (lr, weight_decay, grad_scale) last written to the device state
        self.grad_divisor = None      # device float tensor (1 element): gradients are divided by it inside the kernel
        self._bound_divisor = None
        self.opt_ptr = load_hip_function("nnhipCreateFusedOptimizer")()
        if not self.opt_ptr:
            raise RuntimeError("nnhipCreateFusedOptimizer failed")
        n = len(self.params)
        self.c_params = (c_void_p * n)()
        self.c_grads = (c_void_p * n)()
        self.c_exp_avgs = (c_void_p * n)()
        self.c_exp_avg_sqs = (c_void_p * n)()
        self.c_sizes = (c_int64 * n)()
        self._in_backward = False     # fuse_backward(): the layers' backward kernels apply the update themselves when they can
        self._stepped_in_backward = False
        # the parameters know their optimizer (weakly): a backward kernel that produces ALL of its gradients may wait for
        # step() and take the update into its own launch (experimental/linear.py: _PendingMLPBackward) -- the default path
        self._ref = weakref.ref(self)
        self._index = {id(p): i for i, p in enumerate(self.params)}
        for p in self.params:
            p._opt_ref = self._ref

    def fuse_backward(self, enable: bool = True):
        """"Optimizer in backward" (extension; single process only): a backward kernel that produces ALL of this optimizer's
        gradients may apply the Adam update in its own epilogue -- today the README quick-start MLP's one-launch backward
        (experimental/linear.py: _mlp_chain_backward -> nnhipLinearReLULinearBackwardAdam).  step() then launches nothing for
        that iteration.  Same arithmetic, same results; call step() after every backward() (no gradient accumulation across
        backward passes, no all-reduce between them -- a GradBucket with hooks keeps the ordinary path)."""
        self._in_backward = bool(enable)
        for i, p in enumerate(self.params):
            if enable:
                p._fused_opt = (self, i)
            elif hasattr(p, "_fused_opt"):
                del p._fused_opt

    def can_fuse_into_backward(self):
        """Nothing stands between this optimizer's gradients and its update: no gradient divisor (clipping), no live
        collectives (data parallel: the gradients have to meet the other ranks' first)."""
        if self.grad_divisor is not None:
            return False
        from .distributed import collectives_live
        return not collectives_live()

    def _update_table(self, params, step):
        """(optimizer handle, pointer table {p, m, v} x params, hyper-parameters, step) for a backward kernel that applies the
        update itself; None when `params` are not exactly this optimizer's parameters or one of them is not contiguous."""
        if len(params) != len(self.params) or len({id(p) for p in params}) != len(params):
            return None
        table = (c_void_p * (3 * len(params)))()
        for k, p in enumerate(params):
            i = self._index.get(id(p))
            if i is None or self.params[i] is not p or not p.data.is_contiguous():
                return None
            table[3 * k], table[3 * k + 1], table[3 * k + 2] = p.data.data_ptr(), self.m[i].data_ptr(), self.v[i].data_ptr()
        # the fused kernel reads the divisor from the library handle: a divisor bound by an earlier ordinary step() (clipping
        # switched on for a few steps, then back to None) must be unbound first -- it may point at freed memory by now
        if self._bound_divisor is not None:
            call_hip_function("nnhipFusedOptimizerSetGradDivisor", self.opt_ptr, None)
            self._bound_divisor = None
        if self.device_step:
            self.sync_device_hyper()
        return (self.opt_ptr, table, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                0 if self.device_step else step, self.decay_mode, self.grad_scale)

    def backward_update_args(self, params):
        """fuse_backward(True): for a backward kernel launched INSIDE backward() (step() has not been called yet: step t + 1)."""
        if not self._in_backward or self._stepped_in_backward or not self.can_fuse_into_backward():
            return None
        if any(getattr(p, "_fused_opt", (None,))[0] is not self for p in params):
            return None
        return self._update_table(params, self.t + 1)

    def pending_update_args(self, params, step):
        """For a backward launch that waited for the optimizer (the default path); `step` = the step number being applied."""
        if not self.can_fuse_into_backward():
            return None
        return self._update_table(params, step)

    def take_pending_backward(self):
        """A backward launch that is still waiting for this optimizer (experimental/linear.py: _PendingMLPBackward): launch it
        NOW with the update inside and let the coming step() only count.  For callers that look at the gradients between
        backward() and step() without changing them (GraphedTrainStep binds them to its bucket) -- without this their first
        look would run the plain backward and step() its own launch.  Returns True when the update has been applied."""
        pend = self.params[0]._pending if self.params else None
        if pend is None or self._stepped_in_backward or not pend.run_with_update(self, self.t + 1):
            return False
        self._stepped_in_backward = True
        bump_param_epoch()
        return True

    def __del__(self):
        ptr = getattr(self, "opt_ptr", None)
        if ptr:
            try:
                load_hip_function("nnhipDestroyFusedOptimizer")(ptr)
            except Exception:
                pass
            self.opt_ptr = None

    def step(self):
        self.t += 1
        bump_param_epoch()        # parameters change in place: deferred Linear outputs of this step are now stale
        if self._stepped_in_backward:          # a fused backward kernel already applied this step's update (fuse_backward)
            self._stepped_in_backward = False
            return
        pend = self.params[0]._pending if self.params else None
        if pend is not None:
            try:
                took = pend.run_with_update(self, self.t)
            except Exception:
                self.t -= 1                    # nothing was applied: the bias correction must not run one step ahead
                raise
            if took:
                return                         # the waiting backward launch took the update with it (one kernel, same values)
        idx = 0
        keep = []        # contiguous copies of strided gradients must outlive the single launch below: a freed block
        #                  could be handed to the next .contiguous() and two table entries would alias one buffer
        for i, p in enumerate(self.params):
            if p.grad is None:  # fused_adamw_multitensor.py:92-96: skip params without a gradient
                continue
            g = p.grad
            if not g.is_contiguous():
                g = g.contiguous()
                keep.append(g)
            if not p.data.is_contiguous():
                p.data = p.data.contiguous()
            self.c_params[idx] = p.data.data_ptr()
            self.c_grads[idx] = g.data_ptr()
            self.c_exp_avgs[idx] = self.m[i].data_ptr()
            self.c_exp_avg_sqs[idx] = self.v[i].data_ptr()
            self.c_sizes[idx] = p.data.numel()
            idx += 1
        if idx == 0:
            return
        div = self.grad_divisor
        if div is not self._bound_divisor:
            call_hip_function("nnhipFusedOptimizerSetGradDivisor", self.opt_ptr, div)
            self._bound_divisor = div
        if self.device_step:
            self.sync_device_hyper()
        call_hip_function("nnhipFusedAdamWMultiTensorStep", self.opt_ptr, idx,
                          ctypes.cast(self.c_params, ctypes.POINTER(c_void_p)),
                          ctypes.cast(self.c_grads, ctypes.POINTER(c_void_p)),
                          ctypes.cast(self.c_exp_avgs, ctypes.POINTER(c_void_p)),
                          ctypes.cast(self.c_exp_avg_sqs, ctypes.POINTER(c_void_p)),
                          ctypes.cast(self.c_sizes, ctypes.POINTER(c_int64)),
                          self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                          0 if self.device_step else self.t, self.decay_mode, self.grad_scale,
                          get_current_stream_ptr())

    def use_device_step(self, enable: bool = True):
        """Move the step counter AND lr / weight_decay / grad_scale to device memory (needed before capturing step()
        into a hipGraph: host values would be frozen into the captured kernel arguments).  While enabled, assign
        `opt.lr`, `opt.weight_decay`, `opt.grad_scale` as usual: `sync_device_hyper()` (called by step() and by
        GraphedTrainStep before every replay) writes changed values to the device with one tiny stream-ordered launch."""
        if enable:
            call_hip_function("nnhipFusedOptimizerSetStep", self.opt_ptr, self.t, get_current_stream_ptr())
            self._dev_hyper = None
            self.device_step = True
            self.sync_device_hyper()
        else:
            self.device_step = False

    def sync_device_hyper(self):
        cur = (float(self.lr), float(self.weight_decay), float(self.grad_scale))
        if self.device_step and cur != self._dev_hyper:
            call_hip_function("nnhipFusedOptimizerSetHyper", self.opt_ptr, cur[0], cur[1], cur[2], get_current_stream_ptr())
            self._dev_hyper = cur

    def zero_grad(self):
        for p in self.params:
            p.grad = None


class AdamW(HIPFusedMultiTensorAdamW):
    """neunet.optim.AdamW (optim.py:39-73) on the fused multi-tensor kernel."""


class Adam(HIPFusedMultiTensorAdamW):
    """neunet.optim.Adam (optim.py:4-37): weight decay is L2-on-gradient; default weight_decay=0."""
    decay_mode = L2_ON_GRAD

    def __init__(self, params, lr: float = 0.01, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        super().__init__(params, lr, betas, eps, weight_decay)


CUDAFusedAdamW, CUDAFusedMultiTensorAdamW = HIPFusedAdamW, HIPFusedMultiTensorAdamW
