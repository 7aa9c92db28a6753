"""HIP-accelerated layers: one-to-one counterparts of neunet/nn/experimental/*
(the reference's CUDA* classes are also exported as aliases of the HIP* ones)."""
from .activations import (CUDAFusedSwishAndMul, CUDASoftmax, CUDASwish, HIPFusedSwishAndMul, HIPReLU,  # noqa: F401
                          HIPSoftmax, HIPSwish)
from .linear import CUDALinear, HIPLinear  # noqa: F401
from .linear_swish import CUDALinearSwish, HIPLinearSwish  # noqa: F401
from .losses import CUDACrossEntropyLoss, HIPCrossEntropyLoss  # noqa: F401
from .rmsnorm import CUDARMSNorm, HIPRMSNorm  # noqa: F401
from .conv2d import HIPConv2d  # noqa: F401
from .attention import HIPMultiHeadAttention  # noqa: F401
from .embedding import HIPDropout, HIPEmbedding, HIPPositionalEncoding  # noqa: F401
from .vision import HIPBatchNorm2d, HIPLeakyReLU, HIPMaxPool2d, HIPMSELoss, HIPSigmoid  # noqa: F401
