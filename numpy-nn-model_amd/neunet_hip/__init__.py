"""neunet_hip -- MI355X (gfx950) backend for the dense hot path of AkiRusProd/numpy-nn-model (`neunet`).

Host side: the reference's Module / Tensor / tape shape (neunet/autograd.py, neunet/nn/modules.py);
device side: hand-written HIP kernels in libneunet_hip.so behind a flat C ABI (include/neunet_hip.h),
bound with ctypes exactly where the reference binds its CUDA .so files
(neunet/nn/experimental/utils.py:64-92).
"""
import numpy as np

from . import nn, optim  # noqa: F401
from ._lib import NeunetHipError, lib_path, load_library  # noqa: F401
from .autograd import Tensor  # noqa: F401

float32, int32, int64 = np.float32, np.int32, np.int64


def tensor(data, requires_grad=True, dtype=np.float32, device="cpu"):
    """neunet.tensor (neunet/__init__.py:40-47)."""
    return Tensor(data, requires_grad=requires_grad, dtype=dtype, device=device)


def argmax(x: Tensor, axis=None, keepdims=False):
    """neunet.argmax -> int32 (neunet/__init__.py:132-139); index results must be bit-exact."""
    if x.device == "cpu":
        out = np.argmax(x.data, axis=axis, keepdims=keepdims).astype(np.int32)
        return Tensor(out, dtype=np.int32, requires_grad=False, device="cpu")
    import torch
    from ._lib import call_hip_function, get_current_stream_ptr
    d = x.data
    if d.dtype != torch.float32:
        # np.argmax takes labels / ids / masks too (neunet/__init__.py:132-139).  The library's kernel compares fp32: integer types
        # whose EVERY value is exact in fp32 (|v| < 2^24: bool, uint8, int16) ride on it; wider integers and float64 keep torch's
        # argmax -- same first-maximum rule.  Decided from the dtype alone: a range test on the data (round 5: `d.abs().max()`)
        # is a host read in the middle of a step and breaks hipGraph capture (advisor, round 5).
        if d.dtype in (torch.int16, torch.uint8, torch.bool):
            d = d.to(torch.float32)
        else:
            out = torch.argmax(d) if axis is None else torch.argmax(d, dim=axis, keepdim=keepdims)
            if axis is None and keepdims:
                out = out.reshape((1,) * d.dim())
            return Tensor(out.to(torch.int32), dtype=np.int32, requires_grad=False, device="cuda")
    d = d.contiguous()
    shape = tuple(d.shape)
    if axis is None:
        outer, n, inner, oshape = 1, d.numel(), 1, ((1,) * len(shape) if keepdims else ())
    else:
        ax = axis + len(shape) if axis < 0 else axis
        if not 0 <= ax < len(shape):
            raise ValueError(f"axis {axis} is out of bounds for array of dimension {len(shape)}")
        outer, n, inner = int(np.prod(shape[:ax], dtype=np.int64)), shape[ax], int(np.prod(shape[ax + 1:], dtype=np.int64))
        oshape = shape[:ax] + ((1,) if keepdims else ()) + shape[ax + 1:]
    if n == 0 and outer * inner > 0:
        raise ValueError("attempt to get argmax of an empty sequence")
    out = torch.empty(oshape, dtype=torch.int32, device=d.device)
    call_hip_function("nnhipArgmaxF32", out, d, outer, n, inner, get_current_stream_ptr())
    return Tensor(out, dtype=np.int32, requires_grad=False, device="cuda")


def save(obj, path):
    """neunet.save = pickle (neunet/__init__.py:26-29)."""
    import pickle
    with open(path, "wb") as f:
        pickle.dump(obj, f)


def load(path):
    import pickle
    with open(path, "rb") as f:
        return pickle.load(f)
