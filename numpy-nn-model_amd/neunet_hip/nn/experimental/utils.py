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
    """local_grad * grad for a loss node whose backward() was seeded with something other than ones (cross_entropy.py:111-114).
    The broadcast is decided from grad's SHAPE, as NumPy decides it: a scalar (0-d / one element) and one value per row -- shape
    (rows,) against a 1-D loss, or (rows, 1, ...) -- go through the library's nnhipScaleRows; every other broadcastable shape
    (a (D,) gradient against a (B, D) 'none' loss runs along the LAST axis, also when B == D) is expanded and multiplied
    elementwise (nnhipMul); a shape that does not broadcast raises like NumPy."""
    import torch
    lg = local_grad if local_grad.is_contiguous() else local_grad.contiguous()
    g = grad if isinstance(grad, torch.Tensor) else torch.as_tensor(grad, dtype=torch.float32, device=lg.device)
    g = g.to(torch.float32)
    out = torch.empty_like(lg)
    rows = lg.shape[0] if lg.ndim >= 1 else 1
    cols = lg.numel() // max(rows, 1)
    per_row = (lg.ndim >= 1 and g.ndim == lg.ndim and g.shape[0] == rows and all(d == 1 for d in g.shape[1:])) or \
              (lg.ndim == 1 and tuple(g.shape) == (rows,))
    if g.numel() == 1:
        call_hip_function("nnhipScaleRows", out, lg, g.reshape(1).contiguous(), rows, cols, 0, get_current_stream_ptr())
    elif per_row:
        call_hip_function("nnhipScaleRows", out, lg, g.reshape(rows).contiguous(), rows, cols, 1, get_current_stream_ptr())
    else:
        try:
            full = torch.broadcast_to(g, lg.shape).contiguous()
        except RuntimeError:
            raise ValueError(f"upstream gradient of shape {tuple(g.shape)} does not broadcast to a loss of shape {tuple(lg.shape)}") from None
        call_hip_function("nnhipMul", out, lg, full, lg.numel(), get_current_stream_ptr())
    return out
