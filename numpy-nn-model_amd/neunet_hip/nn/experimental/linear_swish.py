"""HIPLinearSwish -- drop-in for CUDALinearSwish
(neunet/nn/experimental/linear_swish/linear_swish_cutlass.py:198-278).

y = swish(X W^T + b) with the bias add and Swish fused into the MFMA GEMM epilogue (optionally also
writing the pre-activation z).  fp32 MFMA throughout: unlike the reference's TF32 tensor-op path
(linear_swish_cutlass_evt_full.cu:440) this meets the 1e-4 parity target."""
import os
import weakref
from typing import Union

import numpy as np

from ...autograd import Tensor
from ..modules import Module
from ..parameter import Parameter
from .linear import _finish_param, _grad_out, hip_linear_module_backward
from .utils import call_hip_function, get_current_stream_ptr


def hip_linear_swish_forward(X, weights, bias, O, preactivation, input_rows, input_cols, output_cols,
                             swish_beta=1.0, save_preactivation=False):
    """cuda_linear_swish_forward (linear_swish_cutlass.py:68-98)."""
    return call_hip_function("nnhipLinearSwishForward", X, weights, bias, O, preactivation, input_rows,
                             input_cols, output_cols, float(swish_beta), int(save_preactivation),
                             get_current_stream_ptr())


def hip_linear_swish_backward(X, weights, bias, grad_O, d_linear_tmp, grad_X, grad_weight, grad_bias,
                              input_rows, input_cols, output_cols, swish_beta=1.0,
                              recompute_preactivation=True):
    """cuda_linear_swish_backward (linear_swish_cutlass.py:101-139); d_linear_tmp is overwritten with dZ."""
    return call_hip_function("nnhipLinearSwishBackward", X, weights, bias, grad_O, d_linear_tmp, grad_X,
                             grad_weight, grad_bias, input_rows, input_cols, output_cols, float(swish_beta),
                             int(recompute_preactivation), get_current_stream_ptr())


# What `save_preactivation=True` keeps for the backward pass: the pre-activation z (the reference's contract,
# linear_swish_cutlass.py:198-278) or -- the default here since round 6 -- swish'(z), which the forward epilogue gets from the
# sigmoid it computes anyway.  Nothing but the Swish backward ever reads the saved tensor, and with the derivative in hand that
# backward is one multiply: the sigmoid + polynomial it replaces cost ~6 % of the 16384 x 2048 x 512 input-gradient GEMM whose
# epilogue carries it (fp32 MFMA and the vector ALU share lanes).  NNHIP_SWISH_SAVE_DERIVATIVE=0 keeps z.
_SAVE_DERIVATIVE = os.environ.get("NNHIP_SWISH_SAVE_DERIVATIVE", "1") != "0"


class _HIPLinearSwishTensor(Tensor):
    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)
        self_ref = weakref.ref(self)   # the closure must not own the tensor (reference cycle -> freed only by the GC)

        def grad_fn(X: Tensor, weight: Tensor, bias, in_rows_num, in_features, out_features, swish_beta,
                    preactivation, save_preactivation, grad):
            grad = grad if grad.is_contiguous() else grad.contiguous()
            grad_X = X.xp.empty_like(X.data, dtype=np.float32) if X.requires_grad else None
            grad_weight = _grad_out(weight, weight.data)
            grad_bias = _grad_out(bias, bias.data) if bias is not None else None
            me = self_ref()
            if getattr(me, "_grad_is_dz", False):
                # the consumer (a HIPLinear) already applied swish'(z) in its dX epilogue: grad IS dz (linear.py)
                me._grad_is_dz = False
                hip_linear_module_backward(X.data, weight.data, grad, grad_X, grad_weight, grad_bias, in_rows_num,
                                           in_features, out_features)
            else:
                if save_preactivation:
                    d_linear_tmp, recompute = preactivation, (2 if save_preactivation == 2 else 0)
                else:
                    d_linear_tmp, recompute = X.xp.empty((in_rows_num, out_features), dtype=np.float32), True
                hip_linear_swish_backward(X.data, weight.data, bias.data if bias is not None else None, grad,
                                          d_linear_tmp, grad_X, grad_weight, grad_bias, in_rows_num, in_features,
                                          out_features, swish_beta, recompute)
            if grad_X is not None:
                X.apply_grad(grad_X)
            _finish_param(weight, grad_weight)
            if bias is not None:
                _finish_param(bias, grad_bias)

        self.grad_fn = grad_fn


class HIPLinearSwish(Module):
    def __init__(self, in_features, out_features, bias: bool = True, swish_beta: float = 1.0,
                 save_preactivation: bool = True, device="cuda"):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.swish_beta = swish_beta
        self.save_preactivation = save_preactivation
        stdv = 1.0 / np.sqrt(in_features)
        self.weight = Parameter(Tensor(np.random.uniform(-stdv, stdv, (out_features, in_features)), dtype=np.float32))
        self.bias: Union[Tensor, None] = Parameter(
            Tensor(np.random.uniform(-stdv, stdv, (1, out_features)), dtype=np.float32)) if bias else None
        self.to(device)

    def forward(self, X: Tensor) -> Tensor:
        if X.device != self.device:
            raise ValueError(f"Input tensor must be on {self.device}")
        if X.device != "cuda":
            raise NotImplementedError("HIPLinearSwish runs on the HIP device only (no CPU fallback)")
        if X.dtype != "float32":
            raise NotImplementedError(f"Only float32 is supported, got {X.dtype} instead.")
        if not X.data.is_contiguous():
            raise ValueError("HIPLinearSwish needs a C-contiguous input")
        out_shape = X.shape[:-1] + (self.out_features,)
        output = X.xp.empty(out_shape, dtype=np.float32)
        preact = X.xp.empty(out_shape, dtype=np.float32) if self.save_preactivation else None
        rows = int(np.prod(X.shape[:-1]))
        # 0: nothing saved (z is recomputed by the backward GEMM); 1: z; 2: swish'(z) in z's place
        mode = (2 if _SAVE_DERIVATIVE else 1) if self.save_preactivation else 0
        hip_linear_swish_forward(X.data, self.weight.data, self.bias.data if self.bias is not None else None,
                                 output, preact, rows, self.in_features, self.out_features, self.swish_beta, mode)
        return _HIPLinearSwishTensor(output, (X, self.weight, self.bias, rows, self.in_features, self.out_features,
                                              self.swish_beta, preact, mode),
                                     "linear_swish", device=self.device)


CUDALinearSwish = HIPLinearSwish
