"""HIPCrossEntropyLoss -- drop-in for CUDACrossEntropyLoss
(neunet/nn/experimental/losses/cross_entropy_loss/cross_entropy.py:35-150).

One kernel computes per-row loss, log-sum-exp AND d(logits).  Differences from the reference's GPU
path, all on the side of the CPU semantics (neunet/nn/losses.py:59-126):
  * out-of-place by default: `logits` survive the forward (the reference kernel overwrites them,
    cross_entropy.py:69,94); `inplace=True` restores the reference's memory behaviour;
  * ignored rows are excluded from the 'mean' denominator on the DEVICE (nnhipCountNotEqual), so the
    forward never synchronises the host (the reference calls `.item()`, cross_entropy.py:72);
  * the 'mean' / 'sum' reduction also runs on the device (nnhipReduceLoss).
"""
import weakref
from ...autograd import Tensor
from ..modules import Module
from .utils import call_hip_function, contiguous, get_current_stream_ptr, times_upstream

_RED = {"none": b"n", "mean": b"m", "sum": b"s"}


def cross_entropy_forward_backward(logits, labels, reduction: str = "none", ignore_index: int = -100,
                                   inplace: bool = False, weight=None):
    """cross_entropy.py:35-103.  logits (rows, C) f32 device array, labels (rows,) int16/int32/int64 device array
    (the reference's CUDA path takes int32 only, cross_entropy.py:57; its CPU path all three, losses.py:100),
    weight: optional (C,) f32 device array of class weights (losses.py:93-118).
    Returns (loss, grad_logits): loss is a 0-d device array for 'mean'/'sum', (rows,) for 'none'.
    One launch whatever the reduction (nnhipCrossEntropyLossEx)."""
    import torch
    if logits.ndim != 2:
        raise ValueError("Logits must be 2D tensor")
    if labels.ndim != 1:
        raise ValueError("Labels must be 1D tensor")
    if labels.shape[0] != logits.shape[0]:
        raise ValueError("Logits and labels must have the same number of samples")
    if labels.dtype not in _LABEL_BYTES():
        raise TypeError("Labels must be of int16, int32 or int64 dtype")
    if logits.dtype != torch.float32:
        raise TypeError("Logits must be of float32 dtype")
    if reduction not in _RED:
        raise ValueError("Reduction must be 'none', 'mean', or 'sum'")
    rows, vocab = logits.shape
    if weight is not None:
        if tuple(weight.shape) != (vocab,):
            raise ValueError("Weight shape must be equal to number of classes")
        if weight.dtype != torch.float32:
            raise TypeError("Weight must be of float32 dtype")
        weight = contiguous(weight)
    logits, labels = contiguous(logits), contiguous(labels)
    loss_rows = torch.empty(rows, dtype=torch.float32, device=logits.device)
    lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
    grad_logits = logits if inplace else torch.empty_like(logits)
    loss = count = None
    if reduction != "none":
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        count = torch.empty(1, dtype=torch.int32, device=logits.device) if reduction == "mean" else None
    call_hip_function("nnhipCrossEntropyLossEx", logits, None if inplace else grad_logits, loss_rows, lse, labels,
                      _LABEL_BYTES()[labels.dtype], weight, logits.stride(0), int(ignore_index), rows, vocab,
                      _RED[reduction], loss, count, get_current_stream_ptr())
    return (loss_rows if reduction == "none" else loss), grad_logits


def _LABEL_BYTES():
    import torch
    return {torch.int16: 2, torch.int32: 4, torch.int64: 8}


def _fused_linear_cross_entropy(y_pred, labels, reduction, ignore_index, weight):
    """If y_pred is the still-pending output of a small HIPLinear (a classifier head: <= 256 rows, <= 32 classes), run the
    GEMM and the loss as ONE launch (nnhipLinearCrossEntropyLoss) and hand the logits to the Linear's tensor.  Returns
    (loss, grad_logits) or None when the ordinary two-launch path has to run."""
    import torch
    ops = getattr(y_pred, "_operands", None)
    if ops is None or not getattr(y_pred, "pending", lambda: False)() or len(y_pred.shape) != 2:
        return None
    rows, classes = y_pred.shape
    xdata, w, b = ops
    in_features = w.shape[1]
    if not (1 <= rows <= 256 and 1 <= classes <= 32 and 1 <= in_features <= 2048):
        return None
    if labels.ndim != 1 or labels.shape[0] != rows or labels.dtype not in _LABEL_BYTES() or reduction not in _RED:
        return None                                    # let the ordinary path raise its errors
    if weight is not None and (tuple(weight.shape) != (classes,) or weight.dtype != torch.float32):
        return None
    labels = contiguous(labels)
    dev = xdata.device
    logits = torch.empty((rows, classes), dtype=torch.float32, device=dev)
    grad_logits = torch.empty_like(logits)
    loss_rows = torch.empty(rows, dtype=torch.float32, device=dev)
    lse = torch.empty(rows, dtype=torch.float32, device=dev)
    loss = count = None
    if reduction != "none":
        loss = torch.empty((), dtype=torch.float32, device=dev)
        count = torch.empty(1, dtype=torch.int32, device=dev) if reduction == "mean" else None
    call_hip_function("nnhipLinearCrossEntropyLoss", xdata, w, b, logits, grad_logits, loss_rows, lse, labels,
                      _LABEL_BYTES()[labels.dtype], None if weight is None else contiguous(weight), int(ignore_index), rows,
                      in_features, classes, _RED[reduction], loss, count, get_current_stream_ptr())
    y_pred.adopt(logits)
    return (loss_rows if reduction == "none" else loss), grad_logits


class _HIPCrossEntropyTensor(Tensor):
    _implicit_seed = True      # backward() with no argument needs no ones tensor: grad_fn below handles the unit seed

    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)
        out_ref = weakref.ref(self)   # no tensor -> grad_fn -> closure -> tensor cycle: activations die by refcount

        def grad_fn(y_pred: Tensor, grad_y_pred, grad):
            # cross_entropy.py:111-114: y_pred.apply_grad(grad_y_pred * grad).  When backward() was
            # seeded with ones on this very tensor the product is the identity: skip a full pass.
            if getattr(out_ref(), "_seeded_with_ones", False):
                y_pred.apply_grad(grad_y_pred)
                return
            if getattr(grad, "ndim", 0) == 1 and grad.shape[0] != 1:
                grad = grad[:, None]            # cross_entropy.py:112-113: a 1-D upstream gradient is one value per ROW
            y_pred.apply_grad(times_upstream(grad_y_pred, grad))

        self.grad_fn = grad_fn


class HIPCrossEntropyLoss(Module):
    def __init__(self, reduction="none", ignore_index=-100, inplace=False, weight=None):
        """weight (extension over CUDACrossEntropyLoss; neunet.nn.CrossEntropyLoss has it, losses.py:60-64): per-class
        weights as a Tensor / array of shape (C,)."""
        super().__init__()
        self.reduction = reduction
        self.ignore_index = ignore_index
        self.inplace = inplace
        self.weight = None
        if weight is not None:
            import numpy as np
            w = weight.data if isinstance(weight, Tensor) else weight
            self.weight = Tensor(w, dtype=np.float32, requires_grad=False, device="cuda").data

    def forward(self, y_pred: Tensor, y_true: Tensor) -> Tensor:
        if not isinstance(y_pred, Tensor) or not isinstance(y_true, Tensor):
            raise TypeError("Input values must be tensors")
        if y_pred.device != "cuda" or y_true.device != "cuda":
            raise ValueError("Tensors must be on the cuda (HIP) device")
        if y_pred.dtype != "float32":
            raise TypeError("Predictions must be of float32 dtype")
        if y_true.dtype not in ("int16", "int32", "int64"):
            raise TypeError("Target must be of int dtype")
        fused = None if self.inplace else _fused_linear_cross_entropy(y_pred, y_true.data, self.reduction, self.ignore_index,
                                                                     self.weight)
        if fused is not None:
            loss, grad_y_pred = fused
        else:
            loss, grad_y_pred = cross_entropy_forward_backward(y_pred.data, y_true.data, reduction=self.reduction,
                                                               ignore_index=self.ignore_index, inplace=self.inplace,
                                                               weight=self.weight)
        return _HIPCrossEntropyTensor(loss, (y_pred, grad_y_pred), "cross_entropy", device="cuda")


CUDACrossEntropyLoss = HIPCrossEntropyLoss
