"""The remaining modules of the conv-classifier step (examples/convolutional_digits_classifier.ipynb
cell 2): HIPLeakyReLU, HIPSigmoid, HIPMaxPool2d, HIPBatchNorm2d, HIPMSELoss -- same constructor arguments,
`args` layout and gradient formulas as the reference classes they stand in for."""
import ctypes
import os
import weakref
from typing import Union

import numpy as np

from ..._lib import Pool2dDesc, load_hip_function
from ...autograd import Tensor, param_epoch
from ..modules import Module
from ..parameter import Parameter
from .linear import ACT_SIGMOID, _HIPLinearTensor, _finish_param, _grad_out
from .utils import call_hip_function, contiguous, get_current_stream_ptr, require_device_f32, times_upstream


def _pair(v):
    return v if isinstance(v, tuple) else (v, v)


class _Deferred:
    """Mixin (in front of Tensor): `.data` comes from a thunk that runs when somebody first reads it -- unless a consumer
    launched the producing kernel as part of its own and handed the result over with adopt()."""
    _data = None
    _thunk = None
    _lazy_shape = None

    @property
    def data(self):
        if self._data is None and self._thunk is not None:
            thunk, self._thunk = self._thunk, None
            self._data = thunk()
        return self._data

    @data.setter
    def data(self, value):
        self._data = value

    def pending(self) -> bool:
        return self._data is None and self._thunk is not None

    def adopt(self, data):
        self._data, self._thunk = data, None

    @property
    def shape(self):
        return tuple(self._lazy_shape) if self._data is None and self._lazy_shape is not None else tuple(self._data.shape)

    @property
    def dtype(self):
        return np.dtype(np.float32)

    @property
    def ndim(self):
        return len(self.shape)


# ------------------------------------------------------------------------------------------ LeakyReLU / Sigmoid
class _HIPLeakyReLUTensor(Tensor):
    """Output of HIPLeakyReLU.  The elementwise pass is DEFERRED until somebody reads `.data`: a MaxPool2d applied to it
    first (the conv classifier's conv -> LeakyReLU -> MaxPool chain) evaluates the activation inside its pooling window
    and takes over the backward as well (nnhipMaxPool2dLeakyForward / Backward: two launches instead of four per chain);
    anything else that touches `.data` runs the plain kernel at that point.  Values are identical either way."""

    def __init__(self, data, args, op, device, thunk=None, shape=None):
        self._data, self._thunk, self._lazy_shape = None, thunk, shape
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(t: Tensor, out_ref, alpha, grad):
            f_x = out_ref().data                      # materialised by now (a consumer read it) -- or materialise it
            g = t.xp.empty_like(f_x)
            call_hip_function("nnhipLeakyReLUBackward", g, contiguous(grad), f_x, float(alpha), f_x.numel(),
                              get_current_stream_ptr())
            t.apply_grad(g)

        self.grad_fn = grad_fn

    @property
    def data(self):
        if self._data is None and self._thunk is not None:
            thunk, self._thunk = self._thunk, None
            self._data = thunk()
        return self._data

    @data.setter
    def data(self, value):
        self._data = value

    def pending(self) -> bool:
        return self._data is None and self._thunk is not None

    @property
    def shape(self):
        return tuple(self._lazy_shape) if self._data is None and self._lazy_shape is not None else tuple(self._data.shape)

    @property
    def dtype(self):
        return np.dtype(np.float32)

    @property
    def ndim(self):
        return len(self.shape)


# NNHIP_VISION_FUSION=0: no deferred LeakyReLU / MaxPool absorption, no Sigmoid fold into MSE (developer A/B switch)
_FUSE = os.environ.get("NNHIP_VISION_FUSION", "1") != "0"
# NNHIP_BN_HEAD_FUSION=1 (opt-in): the conv + LeakyReLU + MaxPool launch also leaves partial batch statistics and the classifier's
# tail (BatchNorm2d -> flatten -> Linear -> Sigmoid -> MSELoss) runs as ONE launch: C5 in 11 launches instead of 13.  Off by default:
# measured 0.0864 against 0.0853 ms per step (EXPERIMENTS.md, round 5) -- fewer launches, not less time.
_FUSE_TAIL = os.environ.get("NNHIP_BN_HEAD_FUSION", "0") == "1"


class HIPLeakyReLU(Module):
    """neunet/nn/activations.py:72-84."""

    def __init__(self, alpha=0.01):
        super().__init__()
        self.alpha = alpha

    def forward(self, x: Tensor):
        require_device_f32(x)
        alpha = float(self.alpha)

        def thunk():
            f_x = x.xp.empty_like(x.data)
            call_hip_function("nnhipLeakyReLUForward", f_x, contiguous(x.data), alpha, f_x.numel(), get_current_stream_ptr())
            return f_x

        if _FUSE and alpha > 0.0:
            out = _HIPLeakyReLUTensor(None, None, "leakyrelu", device=x.device, thunk=thunk, shape=tuple(x.shape))
        else:
            out = _HIPLeakyReLUTensor(thunk(), None, "leakyrelu", device=x.device)
        out.args = [x, weakref.ref(out), self.alpha]
        return out


class _HIPSigmoidTensor(_Deferred, Tensor):
    def __init__(self, data, args, op, device, thunk=None, shape=None):
        self._thunk, self._lazy_shape = thunk, shape
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(x: Tensor, f_x, grad):
            f_x = self_ref().data if f_x is None else f_x
            g = x.xp.empty_like(f_x)                 # not x.data: x may be a Linear output that was never materialised
            call_hip_function("nnhipSigmoidBackward", g, contiguous(grad), f_x, f_x.numel(), get_current_stream_ptr())
            x.apply_grad(g)

        self_ref = weakref.ref(self)
        self.grad_fn = grad_fn


class HIPSigmoid(Module):
    """neunet/nn/activations.py:19-28."""

    def __init__(self):
        super().__init__()

    def forward(self, x: Tensor):
        require_device_f32(x)
        if isinstance(x, _HIPLinearTensor) and x.pending():
            if _FUSE and getattr(x, "_lazy_input", None) is not None and x._lazy_input.pending():
                # the conv classifier's head on a BatchNorm2d output nobody has looked at yet: stay pending one step longer --
                # HIPMSELoss launches BatchNorm + Linear + Sigmoid + loss as one kernel; any other reader of `.data` gets
                # the Linear + Sigmoid launch below at that moment
                out = _HIPSigmoidTensor(None, [x, None], "sigmoid", device=x.device,
                                        thunk=lambda: x.run_fused(ACT_SIGMOID), shape=tuple(x.shape))
                return out
            f_x = x.run_fused(ACT_SIGMOID)           # sigmoid in the Linear's GEMM epilogue (conv classifier's last layer)
        else:
            f_x = x.xp.empty_like(x.data)
            call_hip_function("nnhipSigmoidForward", f_x, contiguous(x.data), f_x.numel(), get_current_stream_ptr())
        return _HIPSigmoidTensor(f_x, [x, f_x], "sigmoid", device=x.device)


# ----------------------------------------------------------------------------------------------- MaxPool2d
def _pooled_conv_wgrad(X, argmax, desc, pooled, alpha, grad) -> bool:
    """X is the output of a Conv2d whose input needs no gradient and which nobody else consumes (the conv classifier's first
    layer), and the pool's windows tile it exactly: dW, db come straight from the pool's gradient (nnhipConv2dWeightGradPooled) --
    the conv-output gradient is never written, this pool's backward launch and the conv node's own backward are not needed."""
    from .conv2d import _HIPConv2dTensor
    if not _FUSE or not isinstance(X, _HIPConv2dTensor) or getattr(X, "_consumers", 0) != 1 or X.args is None or X.grad is not None:
        return False
    cx, weight, bias, cdesc = X.args
    if cx.requires_grad or not (weight.requires_grad or (bias is not None and bias.requires_grad)):
        return False
    if not call_hip_function("nnhipConv2dWeightGradPooledOk", ctypes.byref(cdesc), ctypes.byref(desc)):
        return False
    from .linear import _finish_param, _grad_out
    grad_W = _grad_out(weight, weight.data)
    grad_b = _grad_out(bias, bias.data) if bias is not None else None
    call_hip_function("nnhipConv2dWeightGradPooled", cx.data, contiguous(grad), argmax, pooled, float(alpha), grad_W, grad_b,
                      ctypes.byref(cdesc), ctypes.byref(desc), get_current_stream_ptr())
    _finish_param(weight, grad_W)
    if bias is not None:
        _finish_param(bias, grad_b)
    X._bwd_done = True                      # the tape skips the conv node (autograd.py: backward)
    return True


class _HIPMaxPool2dTensor(Tensor):
    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(X: Tensor, argmax, desc, pooled, alpha, grad):
            if _pooled_conv_wgrad(X, argmax, desc, pooled, alpha, grad):
                return
            grad_X = X.xp.empty(tuple(X.shape), dtype=np.float32)
            if pooled is None:
                call_hip_function("nnhipMaxPool2dBackward", grad_X, contiguous(grad), argmax, ctypes.byref(desc),
                                  get_current_stream_ptr())
            else:   # X is the INPUT of the LeakyReLU this pool absorbed: route + activation gradient in one launch
                call_hip_function("nnhipMaxPool2dLeakyBackward", grad_X, contiguous(grad), argmax, pooled, float(alpha),
                                  ctypes.byref(desc), get_current_stream_ptr())
            X.apply_grad(grad_X)

        self.grad_fn = grad_fn


def _conv_pool_forward(src, alpha, O, argmax, desc):
    """src is the output of a Conv2d that has not been launched yet and the pool's 2x2 windows tile it: conv, activation and pool
    as one kernel (nnhipConv2dLeakyMaxPoolForward); the conv output stays unwritten unless somebody asks for it later.  Returns
    False (not done), True, or the (stats, blocks, values per block) triple of the statistics variant."""
    from .conv2d import _HIPConv2dTensor
    if not _FUSE or not isinstance(src, _HIPConv2dTensor) or not src.pending():
        return False
    cx, weight, bias, cdesc = src.args
    if not call_hip_function("nnhipConv2dLeakyMaxPoolForwardOk", ctypes.byref(cdesc), ctypes.byref(desc)):
        return False
    nstat = load_hip_function("nnhipConv2dLeakyMaxPoolStatsBlocks")(ctypes.byref(cdesc), ctypes.byref(desc)) if _FUSE_TAIL else 0
    if nstat > 0:
        # the same launch also leaves per-block (mean, M2) pairs of the pooled output, per channel: a BatchNorm2d right behind this
        # pool takes its batch statistics from them (HIPMSELoss: the classifier's whole tail as one launch) -- nothing waits
        stats = O.new_empty((nstat, O.shape[1], 2))
        call_hip_function("nnhipConv2dLeakyMaxPoolForwardStats", cx.data, weight.data, bias.data if bias is not None else None,
                          float(alpha), O, argmax, ctypes.byref(cdesc), ctypes.byref(desc), stats, get_current_stream_ptr())
        return (stats, nstat, (O.shape[0] * O.shape[2] * O.shape[3]) // nstat)
    call_hip_function("nnhipConv2dLeakyMaxPoolForward", cx.data, weight.data, bias.data if bias is not None else None, float(alpha),
                      O, argmax, ctypes.byref(cdesc), ctypes.byref(desc), get_current_stream_ptr())
    return True


class HIPMaxPool2d(Module):
    """neunet/nn/layers/maxpool2d.py:85-249, dilation included (taps at r*dh, s*dw; :170-186)."""

    def __init__(self, kernel_size, stride=None, padding=0, dilation=1):
        super().__init__()
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride) if stride else self.kernel_size
        p = _pair(padding)
        self.padding = (p[0], p[0], p[1], p[1]) if len(p) == 2 else tuple(p)
        self.dilation = _pair(dilation)
        if min(self.dilation) < 1:
            raise ValueError("dilation must be >= 1")

    def forward(self, X: Tensor) -> Tensor:
        import torch
        if not isinstance(X, Tensor):
            raise TypeError("Input must be a tensor")
        require_device_f32(X)
        if X.ndim != 4:
            raise ValueError("MaxPool2d expects a (B, C, H, W) input")
        B, C, H, W = X.shape
        kh, kw = self.kernel_size
        sh, sw = self.stride
        pu, pd, pl, pr = self.padding
        dh, dw = self.dilation
        Ho = (H + pu + pd - dh * (kh - 1) - 1) // sh + 1      # maxpool2d.py:170-183
        Wo = (W + pl + pr - dw * (kw - 1) - 1) // sw + 1
        desc = Pool2dDesc(B, C, H, W, kh, kw, sh, sw, pu, pd, pl, pr, dh, dw)
        O = X.xp.empty((B, C, Ho, Wo), dtype=np.float32)
        argmax = torch.empty((B, C, Ho, Wo), dtype=torch.int32, device=O.device)
        if isinstance(X, _HIPLeakyReLUTensor) and X.pending():
            src, alpha = X.args[0], float(X.args[2])       # pool over LeakyReLU(src) without materialising it
            done = _conv_pool_forward(src, alpha, O, argmax, desc)
            if not done:
                call_hip_function("nnhipMaxPool2dLeakyForward", O, argmax, contiguous(src.data), alpha, ctypes.byref(desc),
                                  get_current_stream_ptr())
            out = _HIPMaxPool2dTensor(O, (src, argmax, desc, O, alpha), "maxpool2d", device=X.device)
            out._chan_stats = done if isinstance(done, tuple) else None
            return out
        done = _conv_pool_forward(X, 1.0, O, argmax, desc)   # pool straight over a conv that has not run yet
        if done:
            out = _HIPMaxPool2dTensor(O, (X, argmax, desc, None, 1.0), "maxpool2d", device=X.device)
            out._chan_stats = done if isinstance(done, tuple) else None
            return out
        call_hip_function("nnhipMaxPool2dForward", O, argmax, contiguous(X.data), ctypes.byref(desc),
                          get_current_stream_ptr())
        return _HIPMaxPool2dTensor(O, (X, argmax, desc, None, 1.0), "maxpool2d", device=X.device)


# --------------------------------------------------------------------------------------------- BatchNorm2d
class _HIPBatchNorm2dTensor(_Deferred, Tensor):
    """Output of HIPBatchNorm2d.  In training mode on an input whose producer has left partial batch statistics (the fused
    conv + LeakyReLU + MaxPool launch, `_chan_stats`) the launch is DEFERRED until somebody reads `.data`: the conv classifier's `bnorm -> reshape -> fc1 -> sigmoid -> MSELoss`
    then runs as ONE kernel (HIPMSELoss, nnhipBatchNorm2dLinearSigmoidMSE); anybody else who touches `.data` gets
    nnhipBatchNorm2dForward at that moment.  The module keeps the pending output referenced and launches it before anything
    can observe the running statistics (the next forward, eval() / train(), a read of running_mean / running_var): a forward
    pass whose result is thrown away still updates them, as the reference's eager layer does (batchnorm2d.py:84-100)."""
    _keeps_pending = True

    def __init__(self, data, args, op, device, thunk=None, shape=None, owner=None):
        self._thunk, self._lazy_shape = thunk, shape
        self._owner = weakref.ref(owner) if owner is not None else None
        self._fusable = None
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(X: Tensor, weight, bias, save_mean, save_inv, affine, grad):
            B, C = X.shape[0], X.shape[1]
            HW = X.shape[2] * X.shape[3]
            grad_X = X.xp.empty_like(X.data)
            gw = _grad_out(weight, weight.data) if affine else None
            gb = _grad_out(bias, bias.data) if affine else None
            call_hip_function("nnhipBatchNorm2dBackward", contiguous(grad), X.data, weight.data if affine else None,
                              save_mean, save_inv, grad_X, gw, gb, B, C, HW, get_current_stream_ptr())
            X.apply_grad(grad_X)
            if affine:
                _finish_param(weight, gw)
                _finish_param(bias, gb)

        self.grad_fn = grad_fn

    def _released(self):
        owner = self._owner() if self._owner is not None else None
        if owner is not None and owner._pending_out is self:
            owner._pending_out = None

    @property
    def data(self):
        if self._data is None and self._thunk is not None:
            thunk, self._thunk = self._thunk, None
            self._data = thunk()
            self._released()
        return self._data

    @data.setter
    def data(self, value):
        self._data = value

    def adopt(self, data):
        self._data, self._thunk = data, None
        self._released()

    def reshape(self, *shape):
        if not self.pending():
            return Tensor.reshape(self, *shape)
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        return _PendingReshape(self, tuple(int(d) for d in shape))


class _PendingReshape(_Deferred, Tensor):
    """x.reshape(...) of an output that is still pending: stays pending with it (the classifier's flatten between BatchNorm2d and
    fc1); same tape node as Tensor.reshape."""
    _keeps_pending = True

    def __init__(self, src, shape):
        total = int(np.prod(src.shape))
        if -1 in shape:
            known = int(np.prod([d for d in shape if d != -1])) or 1
            shape = tuple(total // known if d == -1 else d for d in shape)
        if int(np.prod(shape)) != total:
            raise ValueError(f"cannot reshape a tensor of shape {tuple(src.shape)} into {shape}")
        self.src = src
        self._thunk, self._lazy_shape = (lambda: src.data.reshape(*shape)), shape
        super().__init__(None, (src,), "reshape", device=src.device, requires_grad=src.requires_grad, _nocopy=True)

        def grad_fn(t, grad):
            t.apply_grad(grad.reshape(t.shape))

        self.grad_fn = grad_fn

    def pending(self) -> bool:
        return self._data is None and self.src.pending()


class _RunningStat(Parameter):
    """running_mean / running_var of a HIPBatchNorm2d: reading `.data` first launches a forward pass that is still pending."""
    _stat_owner = None

    @property
    def data(self):
        owner = self._stat_owner() if self._stat_owner is not None else None
        if owner is not None:
            owner._flush_pending()
        return self._stat_data

    @data.setter
    def data(self, value):
        self._stat_data = value


_RunningStat.__name__ = _RunningStat.__qualname__ = "Parameter"     # Module.state_dict / parameters go by the class NAME (modules.py:31)


class HIPBatchNorm2d(Module):
    """neunet/nn/layers/batchnorm2d.py:57-115.  weight/bias/running stats keep the reference's (1, C) shape;
    running_mean / running_var are Parameters with requires_grad=False (in state_dict, not in parameters())."""

    def __init__(self, num_features: int, eps: float = 1e-5, momentum: float = 0.1, affine: bool = True, device="cuda"):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.affine = num_features, eps, momentum, affine
        self.running_mean = Parameter(Tensor(np.zeros((1, num_features)), dtype=np.float32), requires_grad=False)
        self.running_var = Parameter(Tensor(np.ones((1, num_features)), dtype=np.float32), requires_grad=False)
        self.weight: Union[Tensor, None] = Parameter(Tensor(np.ones((1, num_features)), dtype=np.float32)) if affine else None
        self.bias: Union[Tensor, None] = Parameter(Tensor(np.zeros((1, num_features)), dtype=np.float32)) if affine else None
        self.training = True
        self._pending_out = None
        self.to(device)

    def to(self, device):
        self._flush_pending()
        super().to(device)
        me = weakref.ref(self)
        for name in ("running_mean", "running_var"):
            old = self.__dict__[name]
            stat = _RunningStat(Tensor(old.data, dtype=np.float32, device=old.device), requires_grad=False)
            stat._stat_owner = me
            self.__dict__[name] = stat
        return self

    def _flush_pending(self):
        out = self.__dict__.get("_pending_out")
        if out is not None:
            self._pending_out = None
            out.data                                 # launches nnhipBatchNorm2dForward (statistics + running statistics)

    def forward(self, X: Tensor) -> Tensor:
        if not isinstance(X, Tensor):
            raise TypeError("Input must be a tensor")
        if X.device != self.device:
            raise ValueError("Tensors must be on the same device")
        require_device_f32(X)
        if X.ndim != 4 or X.shape[1] != self.num_features:
            raise ValueError("BatchNorm2d expects a (B, C, H, W) input with C == num_features")
        self._flush_pending()
        B, C, H, W = X.shape
        xd = contiguous(X.data)
        save_mean = X.xp.empty((C,), dtype=np.float32)
        save_inv = X.xp.empty((C,), dtype=np.float32)
        w = self.weight.data if self.affine else None
        b = self.bias.data if self.affine else None
        run_mean, run_var = self.running_mean._stat_data, self.running_var._stat_data
        eps, momentum, training = float(self.eps), float(self.momentum), int(bool(self.training))

        def thunk():
            O = X.xp.empty_like(xd)
            call_hip_function("nnhipBatchNorm2dForward", xd, w, b, O, save_mean, save_inv, run_mean, run_var, B, C, H * W, eps,
                              momentum, training, get_current_stream_ptr())
            return O

        args = (X, self.weight, self.bias, save_mean, save_inv, self.affine)
        cstats = getattr(X, "_chan_stats", None)
        if _FUSE and _FUSE_TAIL and training and cstats is not None and X.data is xd:
            out = _HIPBatchNorm2dTensor(None, args, "batchnorm2d", device=self.device, thunk=thunk, shape=(B, C, H, W), owner=self)
            out._fusable = (xd, cstats, w, b, save_mean, save_inv, run_mean, run_var, B, C, H * W, eps, momentum)
            out._epoch = param_epoch()
            self._pending_out = out
            return out
        return _HIPBatchNorm2dTensor(thunk(), args, "batchnorm2d", device=self.device)

    def train(self, mode=True):
        self._flush_pending()
        self.training = mode

    def eval(self):
        self._flush_pending()
        self.training = False


# ------------------------------------------------------------------------------------------------- MSELoss
class _HIPMSETensor(Tensor):
    _implicit_seed = True      # backward() with no argument needs no ones tensor: grad_fn below handles the unit seed

    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)
        out_ref = weakref.ref(self)   # no tensor -> grad_fn -> closure -> tensor cycle: activations die by refcount

        def grad_fn(y_pred: Tensor, grad_pred, grad):
            if getattr(out_ref(), "_seeded_with_ones", False):
                y_pred.apply_grad(grad_pred)
            else:
                y_pred.apply_grad(times_upstream(grad_pred, grad))

        self.grad_fn = grad_fn


def _bn_head_mse(y_pred, y_true):
    """MSELoss(Sigmoid(Linear(BatchNorm2d(x).reshape(B, -1))), target) with every link of the chain still pending: ONE launch
    (nnhipBatchNorm2dLinearSigmoidMSE).  Returns the loss tensor, or None when the pattern / the sizes do not fit (the caller
    reads y_pred.data, which launches the links one by one)."""
    import torch
    if not (isinstance(y_pred, _HIPSigmoidTensor) and y_pred.pending()):
        return None
    lin = y_pred.args[0]
    if not (isinstance(lin, _HIPLinearTensor) and lin.pending() and lin._epoch == param_epoch()):
        return None
    flat = getattr(lin, "_lazy_input", None)
    if not (isinstance(flat, _PendingReshape) and flat.pending()):
        return None
    bn = flat.src
    if not (isinstance(bn, _HIPBatchNorm2dTensor) and bn.pending() and bn._fusable is not None and bn._epoch == param_epoch()):
        return None
    xd, (stats, nstat, count), w, b, save_mean, save_inv, run_mean, run_var, B, C, HW, eps, momentum = bn._fusable
    _, weight, bias, rows, in_f, out_f, _ = lin.args
    if tuple(flat.shape) != (B, C * HW) or rows != B or in_f != C * HW or tuple(y_true.shape) != (B, out_f):
        return None
    if not load_hip_function("nnhipBatchNorm2dLinearSigmoidMSEFits")(B, C, HW, out_f):
        return None
    w_lin, b_lin = lin._weights_now
    if not w_lin.is_contiguous():
        return None
    t = contiguous(y_true.data)
    Y = torch.empty_like(xd)
    pred = torch.empty((B, out_f), dtype=torch.float32, device=xd.device)
    dz = torch.empty_like(pred)
    loss = torch.empty((), dtype=torch.float32, device=xd.device)
    call_hip_function("nnhipBatchNorm2dLinearSigmoidMSE", xd, stats, nstat, count, w, b, Y, save_mean, save_inv, run_mean, run_var, B, C, HW,
                      eps, momentum, w_lin, b_lin, out_f, t, pred, dz, loss, get_current_stream_ptr())
    bn.adopt(Y)
    y_pred.adopt(pred)
    y_pred.args[1] = pred
    return _HIPMSETensor(loss, (lin, dz), "mse", device="cuda")


class HIPMSELoss(Module):
    """neunet/nn/losses.py:9-22 -- sum((pred - true)^2) / numel, loss and d(pred) in one pass."""

    def __init__(self):
        super().__init__()

    def forward(self, y_pred: Tensor, y_true: Tensor) -> Tensor:
        import torch
        if not isinstance(y_pred, Tensor) or not isinstance(y_true, Tensor):
            raise TypeError("Input values must be tensors")
        if y_pred.device != y_true.device:
            raise ValueError("Tensors must be on the same device")
        require_device_f32(y_pred, y_true)
        if y_pred.shape != y_true.shape:
            raise ValueError("MSELoss on the HIP path needs equal shapes")
        fused = _bn_head_mse(y_pred, y_true) if _FUSE else None
        if fused is not None:
            return fused
        p, t = contiguous(y_pred.data), contiguous(y_true.data)
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        dpred = torch.empty_like(p)
        if isinstance(y_pred, _HIPSigmoidTensor) and _FUSE:
            # MSE(Sigmoid(z)): the loss kernel also applies the Sigmoid backward and hands d(loss)/dz straight to z -- the
            # Sigmoid node drops out of this loss's backward (one launch less; other consumers of the Sigmoid output, if
            # any, still go through it and the gradients add up in z)
            call_hip_function("nnhipMSELossSigmoidForwardBackward", p, t, loss, dpred, p.numel(), get_current_stream_ptr())
            return _HIPMSETensor(loss, (y_pred.args[0], dpred), "mse", device="cuda")
        call_hip_function("nnhipMSELossForwardBackward", p, t, loss, dpred, p.numel(), get_current_stream_ptr())
        return _HIPMSETensor(loss, (y_pred, dpred), "mse", device="cuda")
