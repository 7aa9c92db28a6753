"""HIPConv2d -- Conv2d on the implicit-GEMM MFMA kernels (net-new: the reference has no CUDA conv).
CPU semantics: neunet/nn/layers/conv2d.py:120-355 (geometry in Conv2d.build :193-295, forward
:297-355, backward :16-115).  Unlike the reference, weight.data is never mutated (it dilates the
weight in forward and un-dilates it in grad_fn, conv2d.py:307,108 -- SURVEY Appendix A.2)."""
import ctypes
import os
from typing import Union

import numpy as np

from ..._lib import Conv2dDesc
from ...autograd import Tensor, param_epoch
from ..modules import Module
from ..parameter import Parameter
from .linear import _finish_param, _grad_out
from .utils import call_hip_function, get_current_stream_ptr, require_device_f32


def _pair(v):
    return v if isinstance(v, tuple) else (v, v)


def resolve_padding(padding):
    """Conv2d.build's padding normalisation (conv2d.py:196-243) for the forms that are reachable in the
    reference: an int, a (vertical, horizontal) pair, or an (up, down, left, right) 4-tuple.
    (The 'valid'/'same'/'real same' strings are unreachable there: __init__ wraps a str into a pair
    before build() compares it, conv2d.py:164 -> TypeError; they are rejected here.)"""
    if isinstance(padding, str):
        raise ValueError("string paddings are not supported (they raise in the reference as well)")
    p = _pair(padding)
    if len(p) == 2:
        return (p[0], p[0], p[1], p[1])
    if len(p) == 4:
        return tuple(p)
    raise ValueError("padding must be an int, a pair or a 4-tuple")


def conv2d_desc(x_shape, w_shape, stride, padding4, dilation):
    B, Cin, H, W = x_shape
    Cout, Cin_w, kh, kw = w_shape
    if Cin != Cin_w:
        raise ValueError(f"input has {Cin} channels, weight expects {Cin_w}")
    d = Conv2dDesc(B, Cin, H, W, Cout, kh, kw, stride[0], stride[1], dilation[0], dilation[1], *padding4)
    Ho = (H + padding4[0] + padding4[1] - dilation[0] * (kh - 1) - 1) // stride[0] + 1  # conv2d.py:245-258
    Wo = (W + padding4[2] + padding4[3] - dilation[1] * (kw - 1) - 1) // stride[1] + 1
    return d, (Ho, Wo)


def hip_conv2d_forward(X, W, bias, O, desc):
    return call_hip_function("nnhipConv2dForward", X, W, bias, O, ctypes.byref(desc), get_current_stream_ptr())


def hip_conv2d_backward(X, W, grad_O, grad_X, grad_W, grad_b, desc):
    return call_hip_function("nnhipConv2dBackward", X, W, grad_O, grad_X, grad_W, grad_b, ctypes.byref(desc),
                             get_current_stream_ptr())


class _HIPConv2dTensor(Tensor):
    """Output of HIPConv2d.  With a thunk the kernel launch is DEFERRED until somebody reads `.data`: a MaxPool2d(2, 2) applied to
    it first -- directly or through a deferred LeakyReLU -- may run conv, activation and pool as one kernel that never writes the
    conv output (vision.py: nnhipConv2dLeakyMaxPoolForward); anything else that touches `.data` launches the plain forward then.
    Values are identical either way.  The backward never needs the output itself."""

    def __init__(self, data, args, op, device, thunk=None, shape=None):
        self._data, self._thunk, self._lazy_shape = None, thunk, shape
        self._epoch = param_epoch()
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(X: Tensor, weight: Tensor, bias, desc, grad):
            grad = grad if grad.is_contiguous() else grad.contiguous()
            grad_X = X.xp.empty_like(X.data, dtype=np.float32) if X.requires_grad else None
            grad_W = _grad_out(weight, weight.data)
            grad_b = _grad_out(bias, bias.data) if bias is not None else None
            hip_conv2d_backward(X.data, weight.data, grad, grad_X, grad_W, grad_b, desc)
            if grad_X is not None:
                X.apply_grad(grad_X)
            _finish_param(weight, grad_W)
            if bias is not None:
                _finish_param(bias, grad_b)

        self.grad_fn = grad_fn

    @property
    def data(self):
        if self._data is None and self._thunk is not None:
            if self._epoch != param_epoch():
                # the conv would run NOW, on weights an optimizer step has since updated in place: not the forward-time value an
                # eager Conv2d would hold (the same rule as the deferred Linear output, linear.py)
                raise RuntimeError("this Conv2d output was never materialised during its forward pass (the pool that consumed it "
                                   "computed it on the fly) and the parameters have been updated since; read `.data` before "
                                   "optimizer.step(), or set NNHIP_LAZY_CONV=0 for eager Conv2d outputs")
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


_LAZY_CONV = os.environ.get("NNHIP_VISION_FUSION", "1") != "0" and os.environ.get("NNHIP_LAZY_CONV", "1") != "0"


class HIPConv2d(Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=(1, 1), padding=(0, 0),
                 dilation=(1, 1), bias: bool = True, device="cuda"):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.dilation = _pair(stride), _pair(dilation)
        self.padding = resolve_padding(padding)
        stdv = 1.0 / np.sqrt(in_channels * self.kernel_size[0] * self.kernel_size[1])  # conv2d.py:169-183
        self.weight = Parameter(Tensor(
            np.random.uniform(-stdv, stdv, (out_channels, in_channels, *self.kernel_size)), dtype=np.float32))
        self.bias: Union[Tensor, None] = Parameter(Tensor(np.zeros(out_channels), dtype=np.float32)) if bias else None
        self.to(device)

    def forward(self, X: Tensor) -> Tensor:
        if not isinstance(X, Tensor):
            raise TypeError("Input must be a tensor")
        if X.device != self.device:
            raise ValueError("Tensors must be on the same device")
        require_device_f32(X)
        if X.ndim != 4:
            raise ValueError("Conv2d expects a (B, C, H, W) input")
        if not X.data.is_contiguous():
            raise ValueError("HIPConv2d needs a C-contiguous NCHW input")
        desc, (Ho, Wo) = conv2d_desc(X.shape, self.weight.shape, self.stride, self.padding, self.dilation)
        oshape = (X.shape[0], self.out_channels, Ho, Wo)
        weight, bias = self.weight, self.bias

        def thunk():
            O = X.xp.empty(oshape, dtype=np.float32)
            hip_conv2d_forward(X.data, weight.data, bias.data if bias is not None else None, O, desc)
            return O

        if _LAZY_CONV:
            return _HIPConv2dTensor(None, (X, weight, bias, desc), "conv2d", self.device, thunk=thunk, shape=oshape)
        return _HIPConv2dTensor(thunk(), (X, weight, bias, desc), "conv2d", self.device)
