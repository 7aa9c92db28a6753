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
from ...autograd import Tensor
from ..modules import Module
from .utils import call_hip_function, contiguous, get_current_stream_ptr

_RED = {"none": b"n", "mean": b"m", "sum": b"s"}


def cross_entropy_forward_backward(logits, labels, reduction: str = "none", ignore_index: int = -100,
                                   inplace: bool = False):
    """cross_entropy.py:35-103.  logits (rows, C) f32 device array, labels (rows,) int32 device array.
    Returns (loss, grad_logits): loss is a 0-d device array for 'mean'/'sum', (rows,) for 'none'."""
    import torch
    if logits.ndim != 2:
        raise ValueError("Logits must be 2D tensor")
    if labels.ndim != 1:
        raise ValueError("Labels must be 1D tensor")
    if labels.shape[0] != logits.shape[0]:
        raise ValueError("Logits and labels must have the same number of samples")
    if labels.dtype != torch.int32:
        raise TypeError("Labels must be of int32 dtype")
    if logits.dtype != torch.float32:
        raise TypeError("Logits must be of float32 dtype")
    if reduction not in _RED:
        raise ValueError("Reduction must be 'none', 'mean', or 'sum'")
    rows, vocab = logits.shape
    logits, labels = contiguous(logits), contiguous(labels)
    loss_rows = torch.empty(rows, dtype=torch.float32, device=logits.device)
    lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
    grad_logits = logits if inplace else torch.empty_like(logits)
    stream = get_current_stream_ptr()
    if reduction != "none":     # one call: count + rows + reduction (a single launch for small problems)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        count = torch.empty(1, dtype=torch.int32, device=logits.device) if reduction == "mean" else None
        call_hip_function("nnhipCrossEntropyLoss", logits, None if inplace else grad_logits, loss_rows, lse, labels,
                          logits.stride(0), int(ignore_index), rows, vocab, _RED[reduction], loss, count, stream)
        return loss, grad_logits
    count = None
    if reduction == "mean":
        count = torch.empty(1, dtype=torch.int32, device=logits.device)
        call_hip_function("nnhipCountNotEqual", labels, rows, int(ignore_index), count, stream)
    call_hip_function("nnhipCrossEntropyForwardBackward", logits, loss_rows, lse, labels, logits.stride(0),
                      int(ignore_index), rows, vocab, _RED[reduction], -1, count,
                      None if inplace else grad_logits, stream)
    if reduction == "none":
        return loss_rows, grad_logits
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    call_hip_function("nnhipReduceLoss", loss_rows, rows, _RED[reduction], count, loss, stream)
    return loss, grad_logits


class _HIPCrossEntropyTensor(Tensor):
    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)
        out = self

        def grad_fn(y_pred: Tensor, grad_y_pred, grad):
            # cross_entropy.py:111-114: y_pred.apply_grad(grad_y_pred * grad).  When backward() was
            # seeded with ones on this very tensor the product is the identity: skip a full pass.
            if getattr(out, "_seeded_with_ones", False):
                y_pred.apply_grad(grad_y_pred)
                return
            if grad.ndim == 1:
                grad = grad[:, None]
            y_pred.apply_grad(grad_y_pred * grad)

        self.grad_fn = grad_fn


class HIPCrossEntropyLoss(Module):
    def __init__(self, reduction="none", ignore_index=-100, inplace=False):
        super().__init__()
        self.reduction = reduction
        self.ignore_index = ignore_index
        self.inplace = inplace

    def forward(self, y_pred: Tensor, y_true: Tensor) -> Tensor:
        if not isinstance(y_pred, Tensor) or not isinstance(y_true, Tensor):
            raise TypeError("Input values must be tensors")
        if y_pred.device != "cuda" or y_true.device != "cuda":
            raise ValueError("Tensors must be on the cuda (HIP) device")
        if y_pred.dtype != "float32":
            raise TypeError("Predictions must be of float32 dtype")
        if y_true.dtype != "int32":
            raise TypeError("Target must be of int32 dtype")
        loss, grad_y_pred = cross_entropy_forward_backward(y_pred.data, y_true.data, reduction=self.reduction,
                                                           ignore_index=self.ignore_index, inplace=self.inplace)
        return _HIPCrossEntropyTensor(loss, (y_pred, grad_y_pred), "cross_entropy", device="cuda")


CUDACrossEntropyLoss = HIPCrossEntropyLoss
