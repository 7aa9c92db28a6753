"""HIPLinear -- drop-in for CUDALinear (neunet/nn/experimental/linear/linear.py:122-215).

O = X W^T + b on the fp32 MFMA GEMM; backward = two GEMMs + a column sum, one C-ABI call.
N-D inputs are flattened to rows = prod(shape[:-1]) exactly like the reference's CUDA path
(linear.py:193), which equals the CPU path's batched-dW-then-reverse-broadcast result
(neunet/nn/layers/linear.py:17-24 + autograd.py:948-962).
"""
import ctypes
import os
from typing import Union

import numpy as np

from ...autograd import Tensor, bump_param_epoch, param_epoch
from ..modules import Module
from ..parameter import Parameter
from ..._lib import NeunetHipError, load_hip_function, wgrad_flush
from .utils import call_hip_function, get_current_stream_ptr


def hip_linear_module_forward(X, weights, bias, O, input_rows, input_cols, output_cols, addend=None):
    """cuda_linear_module_forward (linear.py:70-92) -> nnhipLinearModuleForward[Ex].
    addend (same shape as O): O = X W^T + b + addend, folded into the GEMM epilogue."""
    if addend is None:
        return call_hip_function("nnhipLinearModuleForward", X, weights, bias, O, input_rows, input_cols,
                                 output_cols, get_current_stream_ptr())
    return call_hip_function("nnhipLinearModuleForwardEx", X, weights, bias, addend, O, input_rows, input_cols,
                             output_cols, get_current_stream_ptr())


def hip_linear_module_backward(X, weights, grad_O, grad_X, grad_weight, grad_bias, input_rows, input_cols,
                               output_cols, grad_X_addend=None):
    """cuda_linear_module_backward (linear.py:95-120) -> nnhipLinearModuleBackward[Ex].
    grad_X / grad_weight / grad_bias may be None (skipped); grad_X_addend: grad_X = dO W + addend."""
    if grad_X_addend is None:
        return call_hip_function("nnhipLinearModuleBackward", X, weights, grad_O, grad_X, grad_weight, grad_bias,
                                 input_rows, input_cols, output_cols, get_current_stream_ptr())
    return call_hip_function("nnhipLinearModuleBackwardEx", X, weights, grad_O, grad_X_addend, grad_X, grad_weight,
                             grad_bias, input_rows, input_cols, output_cols, get_current_stream_ptr())


def _grad_out(param, shape_like):
    """Where a parameter gradient is written: straight into its slot of a flat DP gradient bucket when
    one is attached and this is the first gradient of the step (no pack copy), else a fresh buffer."""
    buf = getattr(param, "_grad_slot", None)
    if buf is not None and param.grad is None:
        return buf
    return param.xp.empty_like(shape_like, dtype=np.float32)


def _plan_fold(X):
    """Can the backward of the activation that produced this Linear's input ride in the dX GEMM's epilogue?
    Returns (act_grad, act_arg, dz, beta) for nnhipLinearModuleBackwardAct / nnhipLinearInputGrad*, or None.
      * act_grad 1 -- X is the output of a fused Linear->Swish that saved its pre-activation z and nothing else consumes it
        (the FFN of examples/gpt.ipynb: fc_2(swish(fc_1(x)))): dz = (dO W) * swish'(z), in place over z -- no separate
        Swish-backward pass over the [rows, d_ff] tensor;
      * act_grad 3 -- the same with the derivative swish'(z) saved by the forward epilogue in z's place (the default since round 6):
        dz = (dO W) * saved, one multiply per element instead of a sigmoid and a polynomial in an MFMA kernel's epilogue;
      * act_grad 2 -- X is the output of a ReLU that nothing else consumes (README quick-start: l2(relu(l1(x)))):
        d(relu input) = (dO W) * [f > 0] -- the separate ReLU-backward launch (at MNIST-MLP scale one more ~5 us graph node)
        goes away."""
    if X.grad is not None or getattr(X, "_consumers", 0) != 1 or not X.requires_grad:
        return None
    if X.op == "linear_swish":
        z, saved = X.args[7], X.args[8]
        if not saved or z is None:
            return None
        # saved == 2: the forward epilogue left swish'(z) there instead of z (linear_swish.py): the fold is a plain multiply
        return (3 if saved == 2 else 1), z, z, float(X.args[6])
    if X.op == "relu":
        f_x = X.args[1]
        return 2, f_x, X.xp.empty_like(f_x, dtype=np.float32), 1.0
    return None


_MLP_CHAIN = os.environ.get("NNHIP_MLP_CHAIN", "1") != "0"


def _mlp_chain_backward(relu_t, w2, b2, grad, rows, hid, out2):
    """Linear2(relu(Linear1(x))) with x needing no gradient (the README quick-start MLP): dW2, db2, dW1, db1 from ONE launch
    (nnhipLinearReLULinearBackward) -- the hidden gradient dZ is formed inside the dW1 tiles and never written, and the ReLU /
    Linear1 tape nodes are marked done.  Returns False when the pattern or the sizes do not fit (caller takes the general path)."""
    if not _LAZY or not _MLP_CHAIN or out2 > 16 or rows > 256:
        return False
    lin1 = relu_t.args[0]
    if not isinstance(lin1, _HIPLinearTensor) or lin1.op != "linear" or lin1.grad is not None or getattr(lin1, "_consumers", 0) != 1:
        return False
    x1, w1, b1, rows1, in1, hid1, residual = lin1.args
    if residual is not None or b1 is None or hid1 != hid or rows1 != rows or (isinstance(x1, Tensor) and x1.requires_grad):
        return False
    if any(getattr(p, "_grad_hook", None) is not None for p in (w1, b1, w2, b2)):
        return False
    if in1 * hid > (1 << 21) or rows * in1 > (1 << 22):       # the small-problem kernels' range (gemm_small_wanted)
        return False
    f_x = relu_t.args[1]
    gw2, gb2 = _grad_out(w2, w2.data), _grad_out(b2, b2.data)
    gw1, gb1 = _grad_out(w1, w1.data), _grad_out(b1, b1.data)
    params, grads = (w2, b2, w1, b1), (gw2, gb2, gw1, gb1)
    clean = all(p_.grad is None for p_ in params)
    # optimizer.fuse_backward(): the same launch also applies Adam to the four parameters (when they are ALL the optimizer has)
    fo = getattr(w2, "_fused_opt", None)
    upd = fo[0].backward_update_args(list(params)) if fo is not None and clean else None
    fits = _mlp_fits(rows, in1, hid, out2)
    if not fits[0]:
        return False
    if upd is None and clean and _AUTO_FUSE_STEP and fits[1] and _auto_fusable(params):
        # The default path (no opt-in): the launch WAITS.  optimizer.step() on exactly these parameters runs it with Adam in its
        # epilogue (one launch instead of two, the same arithmetic); anything that looks at one of the four gradients first
        # -- the user, clipping, a bucket -- runs the plain backward at that moment and step() launches Adam as usual.
        _PendingMLPBackward(x1.data, f_x, w2.data, grad, params, grads, rows, in1, hid, out2, fits[1])
        relu_t._bwd_done = True
        lin1._bwd_done = True
        return True
    try:
        if upd is not None:
            opt_ptr, table, lr, be1, be2, eps, wd, step, mode, gscale = upd
            call_hip_function("nnhipLinearReLULinearBackwardAdam", x1.data, f_x, w2.data, grad, gw2, gb2, gw1, gb1, rows, in1, hid,
                              out2, opt_ptr, ctypes.cast(table, ctypes.POINTER(ctypes.c_void_p)), lr, be1, be2, eps, wd, step, mode,
                              gscale, get_current_stream_ptr())
            fo[0]._stepped_in_backward = True
            # the parameters changed NOW, inside this launch: a deferred output still pending (the logits whose GEMM went into
            # the fused CrossEntropy, a lazy Conv2d output) must raise on a later read instead of recomputing with the updated
            # W2 / b2 -- optimizer.step() bumps again, harmlessly (advisor, round 3)
            bump_param_epoch()
        else:
            call_hip_function("nnhipLinearReLULinearBackward", x1.data, f_x, w2.data, grad, gw2, gb2, gw1, gb1, rows, in1, hid, out2,
                              get_current_stream_ptr())
    except NeunetHipError:                                     # outside the kernel's range: the general path computes the same
        return False
    for p_, g_ in zip(params, grads):
        _finish_param(p_, g_)
    relu_t._bwd_done = True
    lin1._bwd_done = True
    return True


_AUTO_FUSE_STEP = os.environ.get("NNHIP_AUTO_FUSE_STEP", "1") != "0"
_mlp_fits_cache: dict = {}


def _mlp_fits(rows, in1, hid, out2):
    """(plain launch fits, backward + Adam launch fits) on the current device -- nnhipLinearReLULinearBackwardFits, cached."""
    import torch
    key = (rows, in1, hid, out2, torch.cuda.current_device())
    hit = _mlp_fits_cache.get(key)
    if hit is None:
        f = load_hip_function("nnhipLinearReLULinearBackwardFits")
        hit = _mlp_fits_cache[key] = (bool(f(rows, in1, hid, out2, 0)), bool(f(rows, in1, hid, out2, 1)))
    return hit


def _auto_fusable(params):
    """True when one multi-tensor Adam/AdamW owns exactly these parameters and could take the update into the backward launch
    (optim.py: HIPFusedMultiTensorAdamW.pending_update_args decides again at step() time)."""
    ref = getattr(params[0], "_opt_ref", None)
    opt = ref() if ref is not None else None
    if opt is None or len(opt.params) != len(params):
        return False
    if not all(getattr(p_, "_opt_ref", None) is ref for p_ in params) or not opt.can_fuse_into_backward():
        return False
    return True


class _PendingMLPBackward:
    """The README MLP's one-launch backward (nnhipLinearReLULinearBackward), not launched yet.  Attached to the four
    parameters as `_pending` (autograd.Tensor.grad): reading any of their gradients launches the plain backward; the owning
    optimizer's step() launches backward + Adam in one kernel instead (nnhipLinearReLULinearBackwardAdam).  Either way every
    observable value is the one the eager sequence backward(); step() leaves -- only the launch count differs.

    The operands are the forward pass's buffers, held by reference; torch's version counters catch an in-place write to one of
    them between backward() and the launch (a refilled input batch): that raises instead of computing from the wrong data.
    LIMIT of that guard (advisor, round 5): only writes torch knows about bump `_version`.  A buffer rewritten through a raw pointer
    -- one of this library's nnhip* kernels called on `x.data` directly, a captured copy into a static slot during a graph replay
    -- goes unnoticed, and the deferred launch differentiates the new contents.  GraphedTrainStep is safe by construction (it takes
    the pending launch before it binds gradients, inside the same captured step); code that refills operands with library kernels
    between backward() and step() must read a gradient first, or set NNHIP_AUTO_FUSE_STEP=0 (eager backward launch)."""
    __slots__ = ("ops", "versions", "params", "grads", "dims", "live", "adam_fits")

    def __init__(self, x, f_x, w2, grad, params, grads, rows, in1, hid, out2, adam_fits):
        self.adam_fits = adam_fits
        self.ops = (x, f_x, w2, grad)
        self.versions = tuple(getattr(t, "_version", 0) for t in self.ops)
        self.params, self.grads = params, grads
        self.dims = (rows, in1, hid, out2)
        self.live = [True, True, True, True]
        for p_ in params:
            p_._pending = self

    def detach(self, param):
        """`param.grad = value` while pending: the assignment wins over the gradient this pass would have left."""
        for k, p_ in enumerate(self.params):
            if p_ is param:
                self.live[k] = False
        param._pending = None
        if not any(self.live):
            self.ops = None             # nobody is left to see these gradients (backward(); zero_grad()): nothing to launch

    def _release(self):
        for p_ in self.params:
            if p_._pending is self:
                p_._pending = None

    def _check_operands(self):
        if self.ops is None:
            return False
        if tuple(getattr(t, "_version", 0) for t in self.ops) != self.versions:
            self._release()
            raise RuntimeError("a buffer of the forward pass (the input batch, the hidden activation, W2 or the loss gradient) was "
                               "written in place between loss.backward() and the first use of the parameter gradients; the deferred "
                               "backward launch would read the new contents.  Read a gradient (or call optimizer.step()) before "
                               "refilling the buffer, or set NNHIP_AUTO_FUSE_STEP=0 for an eager backward launch")
        return True

    def materialize(self):
        """Somebody reads a gradient: the plain backward, now."""
        if not self._check_operands():
            self._release()
            return
        x, f_x, w2, grad = self.ops
        rows, in1, hid, out2 = self.dims
        gw2, gb2, gw1, gb1 = self.grads
        self._release()                  # before the launch: an error below must not leave the parameters pointing here
        call_hip_function("nnhipLinearReLULinearBackward", x, f_x, w2, grad, gw2, gb2, gw1, gb1, rows, in1, hid, out2,
                          get_current_stream_ptr())
        live, self.live = self.live, [False] * 4
        for p_, g_, on in zip(self.params, self.grads, live):
            if on:
                _finish_param(p_, g_)
        self.ops = None

    def run_with_update(self, opt, step):
        """optimizer.step(): backward + Adam in one launch when `opt` can (all four gradients still wanted, nothing between the
        gradients and the update); returns True when the update has been applied."""
        if not self.adam_fits or not all(self.live) or not self._check_operands():
            return False
        upd = opt.pending_update_args(list(self.params), step)
        if upd is None:
            return False
        x, f_x, w2, grad = self.ops
        rows, in1, hid, out2 = self.dims
        gw2, gb2, gw1, gb1 = self.grads
        opt_ptr, table, lr, be1, be2, eps, wd, step, mode, gscale = upd
        self._release()
        call_hip_function("nnhipLinearReLULinearBackwardAdam", x, f_x, w2, grad, gw2, gb2, gw1, gb1, rows, in1, hid, out2, opt_ptr,
                          ctypes.cast(table, ctypes.POINTER(ctypes.c_void_p)), lr, be1, be2, eps, wd, step, mode, gscale,
                          get_current_stream_ptr())
        self.live = [False] * 4
        for p_, g_ in zip(self.params, self.grads):
            _finish_param(p_, g_)
        self.ops = None
        return True


def _commit_fold(X, dz):
    """Hand dz to the activation's node marked as 'already the gradient of its input'."""
    X.grad = dz.reshape(X.data.shape)
    X._grad_is_dz = True


def _finish_param(param, grad):
    """apply_grad + the DP bucket's gradient-ready hook (GradBucket(overlap=True))."""
    if param.grad is not None:      # accumulation reads `grad` NOW: it must not still be a queued GEMM (_lib.py: wgrad_*)
        wgrad_flush()
    param.apply_grad(grad)
    hook = getattr(param, "_grad_hook", None)
    if hook is not None:
        hook(param)


ACT_SWISH, ACT_RELU, ACT_SIGMOID = 1, 2, 3      # nnhipLinearActivationForward codes
_LAZY = os.environ.get("NNHIP_LAZY_LINEAR", "1") != "0"


class _HIPLinearTensor(Tensor):
    """Output of HIPLinear.  Its GEMM is DEFERRED until somebody reads `.data`: an activation module applied to it
    first (ReLU / Sigmoid / Swish: `act(Linear(x))`, the composition the reference's fused CUDALinearSwish is tested
    against) then launches ONE GEMM with the activation in its epilogue instead of GEMM + elementwise pass; anything
    else that touches `.data` simply runs the plain GEMM at that point.  Values are identical either way."""

    def __init__(self, data, args, op, device, thunk=None, shape=None):
        self._data, self._thunk, self._lazy_shape = None, thunk, shape
        self._epoch = param_epoch()
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(X: Tensor, weight: Tensor, bias, in_rows_num, in_features, out_features, residual, grad):
            grad = grad if grad.is_contiguous() else grad.contiguous()
            if residual is not None:
                residual.apply_grad(grad)       # d(x + linear(h))/dx = 1: the same buffer, by reference
            plan = _plan_fold(X)
            folded = plan is not None
            grad_X = X.xp.empty_like(X.data, dtype=np.float32) if X.requires_grad and not folded else None
            # a gradient X already received (e.g. q/k/v projections sharing one input) is folded into the dX GEMM's
            # epilogue instead of a separate accumulation pass (neunet/autograd.py:85-93 allocates and adds)
            held = X.foldable_grad() if grad_X is not None else None
            grad_weight = _grad_out(weight, weight.data)
            grad_bias = _grad_out(bias, bias.data) if bias is not None else None
            hook = getattr(weight, "_grad_hook", None)
            if folded and hook is None and plan[0] == 2 and bias is not None and _mlp_chain_backward(X, weight, bias, grad,
                                                                                                       in_rows_num, in_features, out_features):
                return
            if folded and hook is None:
                # dz, dW and db from one C call (one LAUNCH for a small layer)
                kind, arg, dz, beta = plan
                call_hip_function("nnhipLinearModuleBackwardAct", X.data, weight.data, grad, arg, kind, beta, dz, grad_weight,
                                  grad_bias, in_rows_num, in_features, out_features, get_current_stream_ptr())
                _commit_fold(X, dz)
                _finish_param(weight, grad_weight)
                if bias is not None:
                    _finish_param(bias, grad_bias)
            elif folded or (hook is not None and grad_X is not None):
                # DP overlap: parameter gradients first, hand them to the bucket (async all-reduce of a finished
                # segment), THEN the input gradient -- the exchange rides under the dX GEMM
                hip_linear_module_backward(X.data, weight.data, grad, None, grad_weight, grad_bias,
                                           in_rows_num, in_features, out_features)
                _finish_param(weight, grad_weight)
                if bias is not None:
                    _finish_param(bias, grad_bias)
                if folded:
                    kind, arg, dz, beta = plan
                    if kind == 1:
                        call_hip_function("nnhipLinearInputGradSwish", grad, weight.data, arg, dz, in_rows_num, in_features,
                                          out_features, beta, get_current_stream_ptr())
                    elif kind == 3:
                        call_hip_function("nnhipLinearInputGradScaled", grad, weight.data, arg, dz, in_rows_num, in_features,
                                          out_features, get_current_stream_ptr())
                    else:
                        call_hip_function("nnhipLinearInputGradReLU", grad, weight.data, arg, dz, in_rows_num, in_features,
                                          out_features, get_current_stream_ptr())
                    _commit_fold(X, dz)
                else:
                    hip_linear_module_backward(X.data, weight.data, grad, grad_X, None, None,
                                               in_rows_num, in_features, out_features, grad_X_addend=held)
            else:
                hip_linear_module_backward(X.data, weight.data, grad, grad_X, grad_weight, grad_bias,
                                           in_rows_num, in_features, out_features, grad_X_addend=held)
                _finish_param(weight, grad_weight)
                if bias is not None:
                    _finish_param(bias, grad_bias)
            if grad_X is not None:
                if held is not None:
                    X.grad = grad_X
                else:
                    X.apply_grad(grad_X)

        self.grad_fn = grad_fn

    # ---- deferred output --------------------------------------------------------------------------------------------
    @property
    def data(self):
        if self._data is None and self._thunk is not None:
            if self._epoch != param_epoch():
                # the GEMM would run NOW, on weights an optimizer step has since updated in place (and possibly on a
                # refilled input buffer): not the forward-time value the reference's eager Linear would hold
                raise RuntimeError("this Linear output was never materialised during its forward pass (its GEMM was fused into "
                                   "the activation / loss that consumed it) and the parameters have been updated since; read "
                                   "`.data` before optimizer.step(), or set NNHIP_LAZY_LINEAR=0 for eager Linear outputs")
            thunk, self._thunk = self._thunk, None
            self._data = thunk(0, 1.0, None)
        return self._data

    @data.setter
    def data(self, value):
        self._data = value

    def pending(self) -> bool:
        return self._data is None and self._thunk is not None

    def adopt(self, data):
        """A consumer launched the deferred GEMM inside its own kernel (losses.py: Linear -> CrossEntropy): `data` is its
        output."""
        self._data, self._thunk, self._operands = data, None, None

    def run_fused(self, activation: int, beta: float = 1.0, save_preactivation: bool = False):
        """Launch the deferred GEMM with `activation` in its epilogue; returns act(XW^T + b).  With save_preactivation the
        same launch also writes z = XW^T + b, which becomes this tensor's data."""
        if save_preactivation:
            z = self.xp.empty(self._lazy_shape, dtype=np.float32)
            out = self._thunk(activation, beta, z)
            self._data, self._thunk = z, None
            return out
        # the pre-activation is not written: the thunk stays so that a second consumer reading `.data` within the same step
        # still gets z = XW^T + b from the forward-time operands (one more GEMM); after a parameter update that read raises
        return self._thunk(activation, beta, None)

    @property
    def shape(self):
        return tuple(self._lazy_shape) if self._data is None and self._lazy_shape is not None else tuple(self._data.shape)

    @property
    def dtype(self):
        return np.dtype(np.float32)

    @property
    def ndim(self):
        return len(self.shape)


class HIPLinear(Module):
    def __init__(self, in_features, out_features, bias: bool = True, device="cuda", backend="mfma"):
        super().__init__()
        if backend not in ("mfma", "cublaslt", "cutlass"):  # the reference's names are accepted and ignored
            raise ValueError(f"Unknown backend: {backend}")
        self.in_features = in_features
        self.out_features = out_features
        self.backend = "mfma"
        stdv = 1.0 / np.sqrt(in_features)  # init as neunet/nn/layers/linear.py:34-45
        self.weight = Parameter(Tensor(np.random.uniform(-stdv, stdv, (out_features, in_features)), dtype=np.float32))
        if bias:
            self.bias: Union[Tensor, None] = Parameter(
                Tensor(np.random.uniform(-stdv, stdv, (1, out_features)), dtype=np.float32))
        else:
            self.bias = None
        self.to(device)

    def forward(self, X: Tensor, residual: Union[Tensor, None] = None) -> Tensor:
        """residual (extension, same shape as the output): returns residual + linear(X) from one kernel -- the
        `x = x + sublayer(...)` of a pre-norm block without the separate add pass."""
        if not isinstance(X, Tensor):
            raise TypeError("Input must be a tensor")
        if X.device != self.device:
            raise ValueError(f"Input tensor must be on {self.device}")
        if X.device != "cuda":
            raise NotImplementedError("HIPLinear runs on the HIP device only (no CPU fallback)")
        if X.dtype != "float32":
            raise NotImplementedError(f"Only float32 is supported, got {X.dtype} instead.")
        if X.shape[-1] != self.in_features:
            raise ValueError(f"Expected last dim {self.in_features}, got {X.shape[-1]}")
        input_rows = int(np.prod(X.shape[:-1]))
        out_shape = tuple(X.shape[:-1]) + (self.out_features,)
        # an input that is itself still pending AND asks to stay so (vision.py: the flattened BatchNorm2d output in front of the
        # conv classifier's head) is not read here: the GEMM's thunk reads it, or a consumer further down launches the whole
        # chain at once (HIPMSELoss: nnhipBatchNorm2dLinearSigmoidMSE)
        lazy_x = residual is None and _LAZY and getattr(X, "_keeps_pending", False) and X.pending()
        xdata = None if lazy_x else (X.data if X.data.is_contiguous() else X.data.contiguous())
        if residual is None and _LAZY:
            weight, bias, in_f, out_f, xp = self.weight, self.bias, self.in_features, self.out_features, X.xp
            w_ptr, b_ptr = weight.data, bias.data if bias is not None else None   # the arrays as they are NOW
            x_src = X

            def launch(activation, beta, preact, xdata=xdata):
                if xdata is None:
                    xdata = x_src.data if x_src.data.is_contiguous() else x_src.data.contiguous()
                out = xp.empty(out_shape, dtype=np.float32)
                if activation == 0:
                    hip_linear_module_forward(xdata, w_ptr, b_ptr, out, input_rows, in_f, out_f)
                elif preact is not None:
                    call_hip_function("nnhipLinearSwishForward", xdata, w_ptr, b_ptr, out, preact, input_rows, in_f, out_f,
                                      float(beta), 1, get_current_stream_ptr())
                else:
                    call_hip_function("nnhipLinearActivationForward", xdata, w_ptr, b_ptr, out, input_rows, in_f, out_f,
                                      activation, float(beta), get_current_stream_ptr())
                return out

            if not lazy_x and xdata is not X.data:
                X = _ContiguousView(X, xdata)
            args = (X, self.weight, self.bias, input_rows, self.in_features, self.out_features, None)
            out = _HIPLinearTensor(None, args, "linear", device=self.device, thunk=launch, shape=out_shape)
            # for a consumer that fuses the pending GEMM into its own launch (None: the input is not there yet)
            out._operands = None if lazy_x else (xdata, w_ptr, b_ptr)
            out._lazy_input = X if lazy_x else None
            out._weights_now = (w_ptr, b_ptr)
            return out
        output = X.xp.empty(out_shape, dtype=np.float32)
        addend = None
        if residual is not None:
            if not isinstance(residual, Tensor) or tuple(residual.shape) != tuple(output.shape) or residual.dtype != "float32":
                raise ValueError("residual must be a float32 tensor of the output's shape")
            addend = residual.data if residual.data.is_contiguous() else residual.data.contiguous()
        hip_linear_module_forward(xdata, self.weight.data, self.bias.data if self.bias is not None else None,
                                  output, input_rows, self.in_features, self.out_features, addend=addend)
        if xdata is not X.data:
            X = _ContiguousView(X, xdata)
        if residual is not None and not residual.requires_grad:
            residual = None
        args = (X, self.weight, self.bias, input_rows, self.in_features, self.out_features, residual)
        out = _HIPLinearTensor(output, args, "linear", device=self.device)
        if residual is not None:
            out.requires_grad = True
        return out


class _ContiguousView(Tensor):
    """A contiguous copy of a non-contiguous parent that forwards its gradient to the parent."""

    def __init__(self, parent: Tensor, data):
        super().__init__(data, (parent,), "contiguous", requires_grad=parent.requires_grad,
                         device=parent.device, _nocopy=True)

        def grad_fn(p, grad):
            p.apply_grad(grad)

        self.grad_fn = grad_fn


CUDALinear = HIPLinear  # drop-in alias for code written against the reference
