"""HIPRMSNorm -- drop-in for CUDARMSNorm (neunet/nn/experimental/rmsnorm/rmsnorm.py:160-221).
CPU semantics: neunet/nn/layers/rmsnorm.py:84-94 (fwd), :43-59 (bwd).

The forward stores only X_std (rows floats); the reference also stores X_norm (a full extra tensor,
rmsnorm.cu:17-113) -- the backward recomputes x/std from X, so the fused forward moves the algorithmic
8 B/elem and the backward 12 B/elem (+ the dw/db partials)."""
from typing import Union

import numpy as np

from ...autograd import Tensor
from ..modules import Module
from ..parameter import Parameter
from .linear import _finish_param, _grad_out
from .utils import call_hip_function, contiguous, get_current_stream_ptr, require_device_f32


def rmsnorm_forward(X, weight, bias, X_norm, X_std, O, eps: float):
    """rmsnorm.py:56-97.  X_norm may be None (not materialised)."""
    if X.shape != O.shape:
        raise ValueError("Input and output shapes must match")
    n_cols = X.shape[-1]
    n_rows = X.numel() // n_cols if n_cols else 0
    call_hip_function("nnhipRMSNormForward", contiguous(X), contiguous(weight),
                      contiguous(bias) if bias is not None else None, O, X_std, X_norm, n_rows, n_cols,
                      float(eps), get_current_stream_ptr())
    return O, X_norm, X_std


def rmsnorm_backward(X, weight, bias, grad_O, grad_X, grad_weight, grad_bias, X_norm, X_std, grad_X_addend=None):
    """rmsnorm.py:100-153.  grad_X_addend (extension): grad_X = rmsnorm gradient + addend in the same pass."""
    if X.shape != grad_O.shape:
        raise ValueError("Input and output shapes must match")
    if grad_X.shape != X.shape:
        raise ValueError("Input and output gradients shapes must match")
    if grad_weight.shape != weight.shape:
        raise ValueError("Weight and weight gradient shapes must match")
    if grad_bias is not None and grad_bias.shape != bias.shape:
        raise ValueError("Bias and bias gradient shapes must match")
    n_cols = X.shape[-1]
    n_rows = X.numel() // n_cols if n_cols else 0
    if grad_X_addend is None:
        call_hip_function("nnhipRMSNormBackward", contiguous(grad_O), contiguous(X), contiguous(weight), X_std,
                          X_norm, grad_X, grad_weight, grad_bias, n_rows, n_cols, get_current_stream_ptr())
    else:
        call_hip_function("nnhipRMSNormBackwardEx", contiguous(grad_O), contiguous(X), contiguous(weight), X_std,
                          X_norm, grad_X_addend, grad_X, grad_weight, grad_bias, n_rows, n_cols,
                          get_current_stream_ptr())
    return grad_X, grad_weight, grad_bias


class _HIPRMSNormTensor(Tensor):
    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(X: Tensor, weight: Tensor, bias, X_norm, X_std, grad):
            grad_X = X.xp.empty_like(X.data, dtype=np.float32)
            grad_weight = _grad_out(weight, weight.data)
            grad_bias = _grad_out(bias, bias.data) if bias is not None else None
            held = X.foldable_grad() if X.requires_grad else None   # e.g. the residual branch's gradient
            rmsnorm_backward(X.data, weight.data, bias.data if bias is not None else None, grad, grad_X,
                             grad_weight, grad_bias, X_norm, X_std, grad_X_addend=held)
            if held is not None:
                X.grad = grad_X
            else:
                X.apply_grad(grad_X)
            _finish_param(weight, grad_weight)
            if bias is not None:
                _finish_param(bias, grad_bias)

        self.grad_fn = grad_fn


class HIPRMSNorm(Module):
    def __init__(self, dim: int, eps: float = 1e-6, device="cuda", bias=False):
        super().__init__()
        self.eps = eps
        self.weight = Parameter(Tensor(np.ones(dim), dtype=np.float32))
        self.bias: Union[Parameter, None] = Parameter(Tensor(np.zeros(dim), dtype=np.float32)) if bias else None
        self.to(device)

    def forward(self, X: Tensor) -> Tensor:
        require_device_f32(X)
        X_std = X.xp.empty(X.shape[:-1], dtype=np.float32)
        O = X.xp.empty_like(X.data)
        rmsnorm_forward(X.data, self.weight.data, self.bias.data if self.bias is not None else None, None,
                        X_std, O, self.eps)
        return _HIPRMSNormTensor(O, (X, self.weight, self.bias, None, X_std), "rmsnorm", device=self.device)


CUDARMSNorm = HIPRMSNorm
