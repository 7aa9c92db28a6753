"""The remaining modules of the conv-classifier step (examples/convolutional_digits_classifier.ipynb
cell 2): HIPLeakyReLU, HIPSigmoid, HIPMaxPool2d, HIPBatchNorm2d, HIPMSELoss -- same constructor arguments,
`args` layout and gradient formulas as the reference classes they stand in for."""
import ctypes
import os
import weakref
from typing import Union

import numpy as np

from ..._lib import Pool2dDesc
from ...autograd import Tensor
from ..modules import Module
from ..parameter import Parameter
from .linear import ACT_SIGMOID, _HIPLinearTensor, _finish_param, _grad_out
from .utils import call_hip_function, contiguous, get_current_stream_ptr, require_device_f32, times_upstream


def _pair(v):
    return v if isinstance(v, tuple) else (v, v)


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


class _HIPSigmoidTensor(Tensor):
    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(x: Tensor, f_x, grad):
            g = x.xp.empty_like(f_x)                 # not x.data: x may be a Linear output that was never materialised
            call_hip_function("nnhipSigmoidBackward", g, contiguous(grad), f_x, f_x.numel(), get_current_stream_ptr())
            x.apply_grad(g)

        self.grad_fn = grad_fn


class HIPSigmoid(Module):
    """neunet/nn/activations.py:19-28."""

    def __init__(self):
        super().__init__()

    def forward(self, x: Tensor):
        require_device_f32(x)
        if isinstance(x, _HIPLinearTensor) and x.pending():
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


def _conv_pool_forward(src, alpha, O, argmax, desc) -> bool:
    """src is the output of a Conv2d that has not been launched yet and the pool's 2x2 windows tile it: conv, activation and pool
    as one kernel (nnhipConv2dLeakyMaxPoolForward); the conv output stays unwritten unless somebody asks for it later."""
    from .conv2d import _HIPConv2dTensor
    if not _FUSE or not isinstance(src, _HIPConv2dTensor) or not src.pending():
        return False
    cx, weight, bias, cdesc = src.args
    if not call_hip_function("nnhipConv2dLeakyMaxPoolForwardOk", ctypes.byref(cdesc), ctypes.byref(desc)):
        return False
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
            if not _conv_pool_forward(src, alpha, O, argmax, desc):
                call_hip_function("nnhipMaxPool2dLeakyForward", O, argmax, contiguous(src.data), alpha, ctypes.byref(desc),
                                  get_current_stream_ptr())
            return _HIPMaxPool2dTensor(O, (src, argmax, desc, O, alpha), "maxpool2d", device=X.device)
        if _conv_pool_forward(X, 1.0, O, argmax, desc):     # pool straight over a conv that has not run yet
            return _HIPMaxPool2dTensor(O, (X, argmax, desc, None, 1.0), "maxpool2d", device=X.device)
        call_hip_function("nnhipMaxPool2dForward", O, argmax, contiguous(X.data), ctypes.byref(desc),
                          get_current_stream_ptr())
        return _HIPMaxPool2dTensor(O, (X, argmax, desc, None, 1.0), "maxpool2d", device=X.device)


# --------------------------------------------------------------------------------------------- BatchNorm2d
class _HIPBatchNorm2dTensor(Tensor):
    def __init__(self, data, args, op, device):
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
        self.to(device)

    def forward(self, X: Tensor) -> Tensor:
        if not isinstance(X, Tensor):
            raise TypeError("Input must be a tensor")
        if X.device != self.device:
            raise ValueError("Tensors must be on the same device")
        require_device_f32(X)
        if X.ndim != 4 or X.shape[1] != self.num_features:
            raise ValueError("BatchNorm2d expects a (B, C, H, W) input with C == num_features")
        B, C, H, W = X.shape
        xd = contiguous(X.data)
        O = X.xp.empty_like(xd)
        save_mean = X.xp.empty((C,), dtype=np.float32)
        save_inv = X.xp.empty((C,), dtype=np.float32)
        call_hip_function("nnhipBatchNorm2dForward", xd, self.weight.data if self.affine else None,
                          self.bias.data if self.affine else None, O, save_mean, save_inv, self.running_mean.data,
                          self.running_var.data, B, C, H * W, float(self.eps), float(self.momentum),
                          int(bool(self.training)), get_current_stream_ptr())
        return _HIPBatchNorm2dTensor(O, (X, self.weight, self.bias, save_mean, save_inv, self.affine), "batchnorm2d",
                                     device=self.device)

    def train(self, mode=True):
        self.training = mode

    def eval(self):
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
