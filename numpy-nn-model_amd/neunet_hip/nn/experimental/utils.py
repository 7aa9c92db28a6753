"""FFI helpers under the same names the reference's experimental layer uses
(neunet/nn/experimental/utils.py:64-92)."""
from ..._lib import (NeunetHipError, call_hip_function, get_current_stream_ptr, load_hip_function,  # noqa: F401
                     load_library, to_pointer)

# drop-in aliases: code written against the reference's names keeps working
call_cuda_function = call_hip_function
load_cuda_function = load_hip_function


def require_device_f32(*tensors, what="input"):
    """The dtype/device checks every reference wrapper performs (e.g. softmax.py:160-164)."""
    for t in tensors:
        if t is None:
            continue
        if t.dtype != "float32":
            raise NotImplementedError(f"Only float32 is supported, got {t.dtype} instead.")
        if t.device != "cuda":
            raise NotImplementedError(f"Only the HIP device ('cuda') is supported, got {t.device} instead.")


def contiguous(a):
    """cp.ascontiguousarray equivalent for device arrays."""
    return a if a.is_contiguous() else a.contiguous()


def times_upstream(local_grad, grad):
    """local_grad * grad for a loss node whose backward() was seeded with something other than ones (cross_entropy.py:111-114):
    `grad` is a device scalar (0-d / 1 element) or one value per row of `local_grad`.  On the library's own kernel
    (nnhipScaleRows); anything else -- a full-shape upstream gradient -- is an elementwise nnhipMul."""
    import torch
    lg = local_grad if local_grad.is_contiguous() else local_grad.contiguous()
    g = grad if isinstance(grad, torch.Tensor) else torch.as_tensor(grad, dtype=torch.float32, device=lg.device)
    g = g.to(torch.float32).contiguous()
    out = torch.empty_like(lg)
    rows = lg.shape[0] if lg.ndim >= 1 else 1
    cols = lg.numel() // max(rows, 1)
    if g.numel() == 1:
        call_hip_function("nnhipScaleRows", out, lg, g.reshape(1), rows, cols, 0, get_current_stream_ptr())
    elif g.numel() == rows:
        call_hip_function("nnhipScaleRows", out, lg, g.reshape(rows), rows, cols, 1, get_current_stream_ptr())
    elif g.numel() == lg.numel():
        call_hip_function("nnhipMul", out, lg, g.reshape(lg.shape), lg.numel(), get_current_stream_ptr())
    else:
        raise ValueError(f"upstream gradient of shape {tuple(g.shape)} does not match a loss of shape {tuple(lg.shape)}")
    return out
