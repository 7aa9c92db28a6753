"""Host-side Tensor + tape: the dispatch hook of the drop-in boundary.

Restates the contract of neunet/autograd.py that the accelerated layers rely on -- and only that:
  * Tensor(data, args, op, requires_grad, dtype, device)           (autograd.py:7-26)
  * apply_grad: reverse-broadcast then assign / accumulate-by-allocation  (autograd.py:85-93, 948-962)
  * backward: seed ones, DFS topological sort over `args`, replay `grad_fn(*args, grad=v.grad)`
    in reverse                                                        (autograd.py:965-1002)
The ~45 elementwise/reduction ops of the reference Tensor are host NumPy glue and out of scope
(SURVEY 2 #1); a few shape ops the hot-path callers need (reshape, scalar mul, add) are provided.

Device arrays: on device "cuda" `Tensor.data` is a torch.Tensor living in HBM (torch-ROCm plays the
role CuPy plays in the reference: allocation + streams only); on "cpu" it is a numpy.ndarray and the
Tensor is only a container -- the dense ops themselves run on "cuda" exclusively (no CPU fallback).
"""
from __future__ import annotations

from typing import Any, Callable, Literal

import numpy as np

from . import _lib

try:  # torch is plumbing: device memory, streams, process groups
    import torch
except Exception:  # pragma: no cover - torch is part of the image
    torch = None

_NP2T = {}
if torch is not None:
    _NP2T = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
             np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64,
             np.dtype(np.int16): torch.int16, np.dtype(np.bool_): torch.bool,
             np.dtype(np.uint8): torch.uint8}
_T2NP = {v: k for k, v in _NP2T.items()}


def _torch_dtype(dt):
    if dt is None:
        return torch.float32
    if torch is not None and isinstance(dt, torch.dtype):
        return dt
    return _NP2T[np.dtype(dt)]


class _DeviceXP:
    """The few array-module calls the accelerated layers make on `X.xp` (the reference uses cupy)."""
    float32 = np.float32
    int32 = np.int32

    @staticmethod
    def empty(shape, dtype=np.float32):
        return torch.empty(tuple(np.atleast_1d(shape).tolist()) if not isinstance(shape, tuple) else shape,
                           dtype=_torch_dtype(dtype), device="cuda")

    @staticmethod
    def zeros(shape, dtype=np.float32):
        return torch.zeros(shape, dtype=_torch_dtype(dtype), device="cuda")

    @staticmethod
    def empty_like(a, dtype=None):
        return torch.empty_like(a, dtype=_torch_dtype(dtype) if dtype is not None else None,
                                memory_format=torch.contiguous_format)

    @staticmethod
    def zeros_like(a, dtype=None):
        return torch.zeros_like(a, dtype=_torch_dtype(dtype) if dtype is not None else None,
                                memory_format=torch.contiguous_format)

    @staticmethod
    def ones_like(a, dtype=None):
        return torch.ones_like(a, dtype=_torch_dtype(dtype) if dtype is not None else None,
                               memory_format=torch.contiguous_format)

    @staticmethod
    def ascontiguousarray(a):
        return a.contiguous()


device_xp = _DeviceXP()


def _to_device_array(data, dtype):
    td = _torch_dtype(dtype)
    if isinstance(data, torch.Tensor):
        return data.to(device="cuda", dtype=td).contiguous().clone() if not data.is_cuda or data.dtype != td \
            else data.contiguous().clone()
    arr = np.ascontiguousarray(np.array(data, dtype=_T2NP[td]))
    return torch.from_numpy(arr).to("cuda")


# Parameter epoch: advanced by every in-place parameter update (optimizer.step(), a graph replay).  A deferred kernel output
# (experimental/linear.py: the lazily launched Linear GEMM) remembers the epoch it was created in; materialising it in a
# LATER epoch would silently compute with the updated weights, so that read raises instead.
_param_epoch = [0]


def param_epoch() -> int:
    return _param_epoch[0]


def bump_param_epoch() -> None:
    _param_epoch[0] += 1


def add_arrays(a, b):
    """apply_grad's `self.grad + grad` (autograd.py:93).  On device: one nnhipAdd launch."""
    if isinstance(a, np.ndarray):
        return a + b
    if a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape \
            and a.is_contiguous() and b.is_contiguous():
        out = torch.empty_like(a)
        _lib.call_hip_function("nnhipAdd", out, a, b, a.numel(), _lib.get_current_stream_ptr())
        return out
    return a + b


class Tensor:
    def __init__(self, data: Any, args=None, op=None, requires_grad: bool = True, dtype=None,
                 device: Literal["cpu", "cuda"] = "cpu", _nocopy: bool = False):
        if device not in ("cpu", "cuda"):
            raise ValueError("Device must be 'cpu' or 'cuda'")
        if isinstance(data, Tensor):
            data = data.data
        if _nocopy:
            # kernel outputs are wrapped, not copied (the reference's `xp.array(data)` in
            # Tensor.__init__ re-copies every op output, autograd.py:15-19)
            self.xp = np if device == "cpu" else device_xp
            self.data = data
        elif device == "cpu":
            self.xp = np
            if torch is not None and isinstance(data, torch.Tensor):
                data = data.detach().cpu().numpy()
            self.data = np.array(data, dtype=dtype if dtype else np.float32)
        else:
            self.xp = device_xp
            self.data = _to_device_array(data, dtype)
        self.grad = None
        self.op = op
        self.args = args
        self.requires_grad = requires_grad
        self.device = device
        self.grad_fn: Callable = lambda *a, **k: None

    # ---- gradient slot ------------------------------------------------------------------------------------------------
    # `grad` is a plain slot except while a backward kernel that produces it is still PENDING (experimental/linear.py:
    # _PendingMLPBackward -- the README MLP's one-launch backward waits for optimizer.step() so that the same launch can
    # apply Adam; whoever reads a gradient first makes the launch happen, so a reader never sees anything but the
    # finished gradient the reference's eager backward would have left, autograd.py:85-93).
    _pending = None

    @property
    def grad(self):
        pend = self._pending
        if pend is not None:
            pend.materialize()
        return self._grad

    @grad.setter
    def grad(self, value):
        pend = self._pending
        if pend is not None:
            pend.detach(self)       # the caller's value replaces what that backward pass would have left here
        self._grad = value

    # ---- wrapping an already-allocated device buffer without a copy (outputs of kernels) --------
    @classmethod
    def _wrap(cls, array, args, op, device, requires_grad=True):
        t = cls.__new__(cls)
        t.xp = np if device == "cpu" else device_xp
        t.data = array
        t.grad = None
        t.op = op
        t.args = args
        t.requires_grad = requires_grad
        t.device = device
        t.grad_fn = lambda *a, **k: None
        return t

    # ---- movement ---------------------------------------------------------------------------------
    def to(self, device):
        if device == self.device:
            return self
        if device not in ("cpu", "cuda"):
            raise ValueError("Device must be 'cpu' or 'cuda'")
        return Tensor(self.data, requires_grad=self.requires_grad, dtype=self.dtype, device=device)

    def cpu(self):
        return self.to("cpu")

    def cuda(self):
        return self.to("cuda")

    def numpy(self) -> np.ndarray:
        """Host copy of the data (the reference's numpy() refuses device tensors; ours copies)."""
        if self.device == "cpu":
            return self.data
        return self.data.detach().cpu().numpy()

    def detach(self):
        return Tensor._wrap(self.data, None, self.op, self.device, requires_grad=False)

    def item(self) -> float:
        return float(self.data.item())

    @property
    def shape(self):
        return tuple(self.data.shape)

    @property
    def dtype(self):
        if isinstance(self.data, np.ndarray):
            return self.data.dtype
        return _T2NP[self.data.dtype]

    @property
    def ndim(self):
        return self.data.ndim

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    # ---- tape ---------------------------------------------------------------------------------------
    def _reverse_broadcast(self, grad):
        """autograd.py:948-962."""
        gshape, sshape = tuple(grad.shape), tuple(self.shape)    # .shape, not .data.shape: a deferred Linear output stays deferred
        if gshape == sshape:
            return grad
        if len(sshape) == grad.ndim:
            axes = tuple(i for i, (a, b) in enumerate(zip(sshape, gshape)) if a != b)
            grad = grad.sum(axes, keepdims=True) if isinstance(grad, np.ndarray) else grad.sum(axes, keepdim=True)
        else:
            padded = (1,) * (grad.ndim - len(sshape)) + sshape
            axes = tuple(i for i, (a, b) in enumerate(zip(padded, gshape)) if a != b)
            grad = grad.sum(axis=axes) if isinstance(grad, np.ndarray) else grad.sum(axes)
        return grad.reshape(sshape)

    def apply_grad(self, grad):
        """autograd.py:85-93: first gradient is kept by reference, later ones accumulate by allocation."""
        if not self.requires_grad:
            return
        grad = self._reverse_broadcast(grad)
        if self.grad is None:
            self.grad = grad
        else:
            self.grad = add_arrays(self.grad, grad)

    def foldable_grad(self):
        """The gradient this tensor already holds, if a kernel can fold it into the next one it produces
        (out = new + held, written to a fresh buffer): same shape, contiguous.  None otherwise."""
        g = self.grad
        if g is None or isinstance(g, np.ndarray) or tuple(g.shape) != tuple(self.shape) or not g.is_contiguous():
            return None
        return g

    def backward(self, grad=None):
        """autograd.py:965-1002."""
        if not self.requires_grad:
            return
        self._seeded_with_ones = grad is None
        if grad is None and getattr(self, "_implicit_seed", False):
            # a loss tensor whose grad_fn already holds d(loss)/d(input) for a unit seed (fused CrossEntropy / MSE):
            # no ones tensor, no fill launch -- its grad_fn ignores `grad` when `_seeded_with_ones` is set
            grad = None
        elif grad is None:
            grad = self.xp.ones_like(self.data, dtype=self.dtype)
        elif self.device == "cpu":
            grad = np.array(grad, dtype=self.dtype)
        else:
            grad = _to_device_array(grad.data if isinstance(grad, Tensor) else grad, self.dtype)
        if grad is not None:
            self.apply_grad(grad)

        tape: list = []
        visited: set = set()
        # iterative post-order DFS (the reference recurses; deep graphs would hit the recursion limit)
        stack = [(self, False)]
        while stack:
            v, expanded = stack.pop()
            if expanded:
                tape.append(v)
                continue
            if id(v) in visited:
                continue
            visited.add(id(v))
            if v.args is None:
                continue
            stack.append((v, True))
            for child in reversed(list(v.args)):
                if isinstance(child, Tensor) and child.requires_grad and id(child) not in visited:
                    stack.append((child, False))
        # how many tape nodes consume each tensor: lets a grad_fn hand a *transformed* gradient to a sole-consumer
        # input (experimental/linear.py folds the Swish backward of a LinearSwish input into its dX GEMM)
        for v in tape:
            for child in v.args:
                if isinstance(child, Tensor):
                    child._consumers = 0
        for v in tape:
            for child in v.args:
                if isinstance(child, Tensor):
                    child._consumers += 1
        # small parameter-gradient GEMMs are queued by the library while the tape is walked and launched a layer's worth at a
        # time (_lib.py: deferred parameter gradients); everything is launched by the time backward() returns
        from ._lib import wgrad_begin, wgrad_end
        deferring = self.device != "cpu" and wgrad_begin()
        try:
            for v in reversed(tape):
                if getattr(v, "_bwd_done", False):      # a consumer's fused backward already produced this node's gradients
                    v._bwd_done = False                 # (experimental/linear.py: Linear2(relu(Linear1(x))) in one launch)
                    continue
                v.grad_fn(*v.args, grad=v.grad)
        finally:
            if deferring:
                wgrad_end()

    # ---- the few shape / scalar ops callers of the hot path need ----------------------------------------
    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        out = Tensor._wrap(self.data.reshape(*shape) if not isinstance(self.data, np.ndarray)
                           else self.data.reshape(shape), (self,), "reshape", self.device, self.requires_grad)

        def grad_fn(t, grad):
            t.apply_grad(grad.reshape(t.shape))

        out.grad_fn = grad_fn
        return out

    def add(self, t: "Tensor") -> "Tensor":
        """Residual add (neunet/autograd.py:96-116): out = self + t, both parents receive the gradient."""
        if not isinstance(t, Tensor):
            raise TypeError("only Tensor + Tensor is provided on the hot path")
        if t.device != self.device:
            raise ValueError("Tensors must be on the same device")
        if tuple(t.shape) != tuple(self.shape):
            raise ValueError("broadcasting adds are host glue and out of scope; shapes must match")
        rg = self.requires_grad or t.requires_grad
        out = Tensor(add_arrays(self.data, t.data), (self, t) if rg else None, "add", requires_grad=rg,
                     device=self.device, _nocopy=True)

        def grad_fn(a, b, grad):
            if a.requires_grad:
                a.apply_grad(grad)
            if b.requires_grad:
                b.apply_grad(grad)

        out.grad_fn = grad_fn
        return out

    __add__ = add

    def __repr__(self):
        return f"Tensor(shape={self.shape}, dtype={self.dtype}, device={self.device}, op={self.op})"
