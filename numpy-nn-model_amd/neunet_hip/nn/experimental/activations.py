"""HIP activations: drop-ins for CUDASwish (experimental/activations/swish/swish.py:92-120),
CUDAFusedSwishAndMul (.../fused_swish_and_mul/fused_swish_and_mul.py:154-179), CUDASoftmax
(.../softmax/softmax.py:139-169), plus HIPReLU (CPU reference: neunet/nn/activations.py:40-59)."""
import weakref

import numpy as np

from ...autograd import Tensor
from ..modules import Module
from .linear import ACT_RELU, ACT_SWISH, _HIPLinearTensor
from .utils import call_hip_function, contiguous, get_current_stream_ptr, require_device_f32


def _is_device_array(a):
    return hasattr(a, "data_ptr") and a.is_cuda


def _check_arrays(*arrs):
    if not all(_is_device_array(a) for a in arrs):
        raise ValueError("All arguments must be device (torch cuda) arrays.")


# ------------------------------------------------------------------------------------------- ReLU
def hip_relu_forward(x, out):
    _check_arrays(x, out)
    call_hip_function("nnhipReLUForward", out, contiguous(x), x.numel(), get_current_stream_ptr())
    return out


def hip_relu_backward(grad_input, grad_output, f_x):
    _check_arrays(grad_input, grad_output, f_x)
    call_hip_function("nnhipReLUBackward", grad_input, contiguous(grad_output), contiguous(f_x),
                      f_x.numel(), get_current_stream_ptr())
    return grad_input


class _HIPReLUTensor(Tensor):
    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)
        self_ref = weakref.ref(self)

        def grad_fn(t: Tensor, f_x, grad):
            me = self_ref()
            if getattr(me, "_grad_is_dz", False):
                # the consumer (a HIPLinear) already applied [f > 0] in its dX epilogue (linear.py:_plan_fold)
                me._grad_is_dz = False
                t.apply_grad(grad)
                return
            grad_input = t.xp.empty_like(f_x)        # not t.data: t may be a Linear output that was never materialised
            hip_relu_backward(grad_input, grad, f_x)
            t.apply_grad(grad_input)

        self.grad_fn = grad_fn


class HIPReLU(Module):
    def __init__(self):
        super().__init__()

    def forward(self, x: Tensor):
        require_device_f32(x)
        if isinstance(x, _HIPLinearTensor) and x.pending():
            f_x = x.run_fused(ACT_RELU)              # relu in the Linear's GEMM epilogue: one launch instead of two
        else:
            f_x = x.xp.empty_like(x.data)
            hip_relu_forward(x.data, f_x)
        return _HIPReLUTensor(f_x, [x, f_x], "relu", device=x.device)


# ------------------------------------------------------------------------------------------ Swish
def hip_swish_forward(x, out, beta: float):
    """cuda_swish_forward (swish.py:40-63)."""
    _check_arrays(x, out)
    if x.shape != out.shape:
        raise ValueError("Input and output shapes must match")
    call_hip_function("nnhipSwishForward", out, contiguous(x), float(beta), x.numel(), get_current_stream_ptr())
    return out


def hip_swish_backward(grad_input, grad_output, x, beta: float):
    """cuda_swish_backward (swish.py:65-90)."""
    _check_arrays(grad_input, grad_output, x)
    if not (grad_input.shape == grad_output.shape == x.shape):
        raise ValueError("Shapes must match")
    call_hip_function("nnhipSwishBackward", grad_input, contiguous(grad_output), contiguous(x), float(beta),
                      x.numel(), get_current_stream_ptr())
    return grad_input


class _HIPSwishTensor(Tensor):
    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(x: Tensor, beta, grad):
            grad_input = x.xp.empty_like(x.data)
            hip_swish_backward(grad_input, grad, x.data, beta)
            x.apply_grad(grad_input)

        self.grad_fn = grad_fn


class HIPSwish(Module):
    def __init__(self, beta: float = 1.0):
        super().__init__()
        self.beta = beta

    def forward(self, x: Tensor):
        require_device_f32(x)
        if isinstance(x, _HIPLinearTensor) and x.pending():
            # Swish(Linear(x)): one GEMM writes z (the Linear's own output, needed by the Swish backward) and swish(z)
            out = x.run_fused(ACT_SWISH, self.beta, save_preactivation=True)
        else:
            out = x.xp.empty_like(x.data)
            hip_swish_forward(x.data, out, self.beta)
        return _HIPSwishTensor(out, [x, self.beta], "swish", device=x.device)


# -------------------------------------------------------------------------------- SwiGLU gate
def hip_fused_swish_and_mul(x, out, beta: float = 1.0, hidden_size=None):
    """cuda_fused_swish_and_mul (fused_swish_and_mul.py:44-89): x rows = [gate | up]."""
    _check_arrays(x, out)
    if x.ndim < 1:
        raise ValueError("Input must have at least 1 dimension.")
    if hidden_size is None:
        hidden_size = out.shape[-1]
    if hidden_size <= 0:
        raise ValueError("hidden_size must be > 0.")
    if x.shape[-1] != hidden_size * 2:
        raise ValueError("Input last dimension must be exactly 2 * hidden_size.")
    if tuple(out.shape) != tuple(x.shape[:-1]) + (hidden_size,):
        raise ValueError("Output shape must be input.shape[:-1] + (hidden_size,).")
    call_hip_function("nnhipFusedSwishAndMul", out, contiguous(x), float(beta), hidden_size, out.numel(),
                      get_current_stream_ptr())
    return out


def hip_fused_swish_and_mul_backward(grad_input, grad_output, x, beta: float = 1.0, hidden_size=None):
    """cuda_fused_swish_and_mul_backward (fused_swish_and_mul.py:92-135)."""
    _check_arrays(grad_input, grad_output, x)
    if hidden_size is None:
        hidden_size = grad_output.shape[-1]
    if hidden_size <= 0:
        raise ValueError("hidden_size must be > 0.")
    if x.shape[-1] != hidden_size * 2:
        raise ValueError("Input last dimension must be exactly 2 * hidden_size.")
    if tuple(grad_output.shape) != tuple(x.shape[:-1]) + (hidden_size,):
        raise ValueError("grad_output shape must be input.shape[:-1] + (hidden_size,).")
    if grad_input.shape != x.shape:
        raise ValueError("grad_input shape must match input shape.")
    call_hip_function("nnhipFusedSwishAndMulBackward", grad_input, contiguous(grad_output), contiguous(x),
                      float(beta), hidden_size, grad_output.numel(), get_current_stream_ptr())
    return grad_input


class _HIPFusedSwishAndMulTensor(Tensor):
    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(x: Tensor, beta: float, grad):
            grad_input = x.xp.empty_like(x.data)
            hip_fused_swish_and_mul_backward(grad_input, grad, x.data, beta=beta)
            x.apply_grad(grad_input)

        self.grad_fn = grad_fn


class HIPFusedSwishAndMul(Module):
    def __init__(self, beta: float = 1.0):
        super().__init__()
        self.beta = beta

    def forward(self, x: Tensor):
        require_device_f32(x)
        if x.ndim < 1:
            raise ValueError("Input must have at least 1 dimension.")
        if x.shape[-1] % 2 != 0:
            raise ValueError("Input last dimension must be divisible by 2.")
        hidden = x.shape[-1] // 2
        out = x.xp.empty(x.shape[:-1] + (hidden,), dtype=np.float32)
        hip_fused_swish_and_mul(x.data, out, beta=self.beta)
        return _HIPFusedSwishAndMulTensor(out, [x, self.beta], "fused_swish_and_mul", device=x.device)


# ---------------------------------------------------------------------------------------- Softmax
def _slices(a, dim):
    dim = dim % a.ndim
    slice_size = a.shape[dim]
    num_slices = a.numel() // slice_size if slice_size else 0
    stride = a.stride(dim)  # elements (softmax.py:83 divides byte strides by itemsize)
    return num_slices, slice_size, stride


def hip_softmax_forward(x, o, dim: int):
    """cuda_softmax_forward (softmax.py:52-94): arbitrary axis through (num_slices, slice_size, stride)."""
    _check_arrays(x, o)
    if x.shape != o.shape:
        raise ValueError("Input and output shapes must match")
    x = contiguous(x)
    n, s, st = _slices(x, dim)
    call_hip_function("nnhipSoftmaxForward", o, x, n, s, st, get_current_stream_ptr())
    return o


def hip_softmax_backward(grad_x, grad, f_x, dim: int):
    """cuda_softmax_backward (softmax.py:96-136)."""
    _check_arrays(grad_x, grad, f_x)
    if grad_x.shape != grad.shape or grad_x.shape != f_x.shape:
        raise ValueError("Input, output gradients and softmax shapes must match")
    grad, f_x = contiguous(grad), contiguous(f_x)
    n, s, st = _slices(f_x, dim)
    call_hip_function("nnhipSoftmaxBackward", grad_x, grad, f_x, n, s, st, get_current_stream_ptr())
    return grad_x


class _HIPSoftmaxTensor(Tensor):
    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(t: Tensor, f_x, axis, grad):
            grad_x = t.xp.empty_like(t.data)
            hip_softmax_backward(grad_x, grad, f_x, axis)
            t.apply_grad(grad_x)

        self.grad_fn = grad_fn


class HIPSoftmax(Module):
    def __init__(self, axis: int = 1):
        super().__init__()
        self.axis = axis

    def forward(self, x: Tensor):
        require_device_f32(x)
        f_x = x.xp.empty_like(x.data)
        hip_softmax_forward(x.data, f_x, self.axis)
        return _HIPSoftmaxTensor(f_x, [x, f_x, self.axis], "softmax", device=x.device)


CUDASwish, CUDAFusedSwishAndMul, CUDASoftmax = HIPSwish, HIPFusedSwishAndMul, HIPSoftmax
