"""neunet_hip.nn -- the names user code takes from `neunet.nn`, bound to the HIP implementations for the
dense hot path (Linear, Conv2d, ReLU, Swish, Softmax, RMSNorm, CrossEntropyLoss)."""
from .modules import Module, ModuleList, Sequential  # noqa: F401
from .parameter import Parameter  # noqa: F401
from . import experimental  # noqa: F401
from .experimental import (HIPConv2d as Conv2d, HIPCrossEntropyLoss, HIPLinear as Linear,  # noqa: F401
                           HIPLinearSwish as LinearSwish, HIPReLU as ReLU, HIPRMSNorm as RMSNorm,
                           HIPSoftmax as Softmax, HIPSwish as Swish, HIPFusedSwishAndMul as FusedSwishAndMul,
                           HIPEmbedding as Embedding, HIPDropout as Dropout, HIPMultiHeadAttention as MultiHeadAttention,
                           HIPPositionalEncoding as PositionalEncoding, HIPBatchNorm2d as BatchNorm2d,
                           HIPLeakyReLU as LeakyReLU, HIPMaxPool2d as MaxPool2d, HIPMSELoss as MSELoss,
                           HIPSigmoid as Sigmoid)


class CrossEntropyLoss(HIPCrossEntropyLoss):
    """neunet.nn.CrossEntropyLoss signature (losses.py:59-64): weight, ignore_index, reduction (default 'mean')."""

    def __init__(self, weight=None, ignore_index=-100, reduction="mean", inplace=False):
        super().__init__(reduction=reduction, ignore_index=ignore_index, inplace=inplace, weight=weight)
