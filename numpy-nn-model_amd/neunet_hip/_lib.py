"""ctypes binding of libneunet_hip.so -- the FFI boundary.

Mirrors neunet/nn/experimental/utils.py:64-92 of the reference (load_cuda_function / to_pointer /
call_cuda_function / get_current_stream_ptr), with two deliberate differences: the library path is
resolved relative to this package (the reference's paths are CWD-relative constants, utils.py:4-62)
and every export returns an int status that is turned into a Python exception here.

There is NO CPU fallback: if the shared library is missing or a launch fails, the call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char, c_double, c_float, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_ENV = "NEUNET_HIP_LIB"
DEFAULT_LIB = os.path.join(_HERE, "lib", "libneunet_hip.so")


class NeunetHipError(RuntimeError):
    """Raised for any non-zero status from libneunet_hip.so."""


class Conv2dDesc(ctypes.Structure):
    """struct nnhipConv2dDesc (include/neunet_hip.h)."""
    _fields_ = [(n, c_int64) for n in ("B", "Cin", "H", "W", "Cout", "kh", "kw", "sh", "sw", "dh", "dw",
                                       "pu", "pd", "pl", "pr")]


class AttentionOptions(ctypes.Structure):
    """struct nnhipAttentionOptions (include/neunet_hip.h)."""
    _fields_ = [("mask_bits", c_void_p), ("mask_bitsT", c_void_p), ("row_any", c_void_p), ("dropout_mask", c_void_p),
                ("dropout_p", c_float), ("dropout_seed", ctypes.c_uint32), ("dropout_seed_dev", c_void_p)]


class Pool2dDesc(ctypes.Structure):
    """struct nnhipPool2dDesc (include/neunet_hip.h)."""
    _fields_ = [(n, c_int64) for n in ("B", "C", "H", "W", "kh", "kw", "sh", "sw", "pu", "pd", "pl", "pr", "dh", "dw")]


P = c_void_p  # device pointers travel as void*
_SIGNATURES = {
    # name: (restype, argtypes)
    "nnhipVersion": (ctypes.c_int, []),
    "nnhipGetLastErrorString": (ctypes.c_char_p, []),
    "nnhipCleanup": (ctypes.c_int, []),
    "nnhipDeviceError": (ctypes.c_int, []),
    "nnhipClearDeviceError": (ctypes.c_int, []),
    "nnhipRaiseDeviceErrorForTest": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p]),
    "nnhipSetGemmMode": (ctypes.c_int, [ctypes.c_int]),
    "nnhipGetGemmMode": (ctypes.c_int, []),
    "nnhipSetGemmLockstep": (ctypes.c_int, [ctypes.c_int]),
    "nnhipGetGemmLockstep": (ctypes.c_int, []),
    "nnhipGemmLaunchCount": (c_int64, [ctypes.c_int]),
    "nnhipWorkspaceReserve": (ctypes.c_int, [c_int64]),
    "nnhipWorkspaceLock": (ctypes.c_int, [ctypes.c_int]),
    "nnhipLinearModuleForward": (ctypes.c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipLinearModuleBackward": (ctypes.c_int, [P, P, P, P, P, P, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipWeightGradDefer": (ctypes.c_int, [ctypes.c_int32, c_void_p]),
    "nnhipWeightGradFlush": (ctypes.c_int, [c_void_p]),
    "nnhipWeightGradFlushGemms": (ctypes.c_int, [c_void_p]),
    "nnhipWeightGradPending": (ctypes.c_int, []),
    "nnhipLinearInputGradSwish": (ctypes.c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_float, c_void_p]),
    "nnhipLinearInputGradScaled": (ctypes.c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipLinearInputGradReLU": (ctypes.c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipLinearReLULinearBackward": (ctypes.c_int, [P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipLinearReLULinearBackwardAdam": (ctypes.c_int, [P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_void_p,
                                                          ctypes.POINTER(c_void_p), ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                                          ctypes.c_double, ctypes.c_double, ctypes.c_int32, ctypes.c_int32, c_float, c_void_p]),
    "nnhipLinearReLULinearBackwardFits": (ctypes.c_int, [c_int64, c_int64, c_int64, c_int64, ctypes.c_int32]),
    "nnhipLinearModuleBackwardAct": (ctypes.c_int, [P, P, P, P, c_int32, c_float, P, P, P, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipLinearActivationForward": (ctypes.c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_int32, c_float, c_void_p]),
    "nnhipLinearModuleForwardEx": (ctypes.c_int, [P, P, P, P, P, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipLinearModuleBackwardEx": (ctypes.c_int, [P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipLinearSwishForward": (ctypes.c_int, [P, P, P, P, P, c_int64, c_int64, c_int64, c_float, ctypes.c_int, c_void_p]),
    "nnhipLinearSwishBackward": (ctypes.c_int, [P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_float, ctypes.c_int, c_void_p]),
    "nnhipGemmF32": (ctypes.c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, ctypes.c_int, ctypes.c_int,
                                    c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipGemmF32Ex": (ctypes.c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, ctypes.c_int, ctypes.c_int,
                                      c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_void_p]),
    "nnhipMaskedSoftmaxForward": (ctypes.c_int, [P, P, P, c_int64, c_int64, c_int64, c_int64, c_float, ctypes.c_int, c_void_p]),
    "nnhipMaskedSoftmaxBackward": (ctypes.c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_float, ctypes.c_int, c_void_p]),
    "nnhipMaskedSoftmaxForwardEx": (ctypes.c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_float, ctypes.c_int, c_void_p]),
    "nnhipMaskedSoftmaxBackwardEx": (ctypes.c_int, [P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_float, ctypes.c_int, c_void_p]),
    "nnhipAttentionForward": (ctypes.c_int, [P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_float, ctypes.c_int, c_void_p]),
    "nnhipAttentionBackward": (ctypes.c_int, [P, P, P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_float, ctypes.c_int, c_void_p]),
    "nnhipAttentionForwardEx": (ctypes.c_int, [P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_float, ctypes.c_int,
                                               POINTER(AttentionOptions), c_void_p]),
    "nnhipAttentionBackwardEx": (ctypes.c_int, [P, P, P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_float,
                                                ctypes.c_int, POINTER(AttentionOptions), c_void_p]),
    "nnhipAttentionPackMask": (ctypes.c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipAttentionDropoutMask": (ctypes.c_int, [P, c_int64, c_int64, c_int64, c_int64, c_float, ctypes.c_uint32, c_void_p]),
    "nnhipAttentionDropoutMaskEx": (ctypes.c_int, [P, c_int64, c_int64, c_int64, c_int64, c_float, ctypes.c_uint32, P, c_void_p]),
    "nnhipEmbeddingForward": (ctypes.c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_int64, c_float, c_void_p]),
    "nnhipEmbeddingBackward": (ctypes.c_int, [P, P, P, c_int64, c_int64, c_int64, c_float, c_void_p]),
    "nnhipNotEqualInt32": (ctypes.c_int, [P, P, c_int64, ctypes.c_int32, c_void_p]),
    "nnhipArgmaxF32": (ctypes.c_int, [P, P, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipDropout": (ctypes.c_int, [P, P, c_int64, c_float, ctypes.c_uint32, P, c_void_p]),
    "nnhipIncrementU32": (ctypes.c_int, [P, ctypes.c_uint32, c_void_p]),
    "nnhipMul": (ctypes.c_int, [P, P, P, c_int64, c_void_p]),
    "nnhipReLUForward": (ctypes.c_int, [P, P, c_int64, c_void_p]),
    "nnhipReLUBackward": (ctypes.c_int, [P, P, P, c_int64, c_void_p]),
    "nnhipSwishForward": (ctypes.c_int, [P, P, c_float, c_int64, c_void_p]),
    "nnhipSwishBackward": (ctypes.c_int, [P, P, P, c_float, c_int64, c_void_p]),
    "nnhipFusedSwishAndMul": (ctypes.c_int, [P, P, c_float, c_int64, c_int64, c_void_p]),
    "nnhipFusedSwishAndMulBackward": (ctypes.c_int, [P, P, P, c_float, c_int64, c_int64, c_void_p]),
    "nnhipSoftmaxForward": (ctypes.c_int, [P, P, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipSoftmaxBackward": (ctypes.c_int, [P, P, P, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipCrossEntropyForwardBackward": (ctypes.c_int, [P, P, P, P, c_int64, c_int32, c_int64, c_int64, c_char, c_int64, P, P, c_void_p]),
    "nnhipCountNotEqual": (ctypes.c_int, [P, c_int64, c_int32, P, c_void_p]),
    "nnhipCrossEntropyDenominator": (ctypes.c_int, [P, c_int32, c_int64, c_int64, P, c_int64, P, P, c_void_p]),
    "nnhipReduceLoss": (ctypes.c_int, [P, c_int64, c_char, P, P, c_void_p]),
    "nnhipCrossEntropyLoss": (ctypes.c_int, [P, P, P, P, P, c_int64, ctypes.c_int32, c_int64, c_int64, ctypes.c_char, P, P, c_void_p]),
    "nnhipLinearCrossEntropyLoss": (ctypes.c_int, [P, P, P, P, P, P, P, P, c_int32, P, c_int64, c_int64, c_int64, c_int64, c_char, P, P,
                                                   c_void_p]),
    "nnhipCrossEntropyLossEx": (ctypes.c_int, [P, P, P, P, P, c_int32, P, c_int64, c_int64, c_int64, c_int64, c_char, P, P, c_void_p]),
    "nnhipRMSNormForward": (ctypes.c_int, [P, P, P, P, P, P, c_int64, c_int64, c_float, c_void_p]),
    "nnhipRMSNormBackward": (ctypes.c_int, [P, P, P, P, P, P, P, P, c_int64, c_int64, c_void_p]),
    "nnhipRMSNormBackwardEx": (ctypes.c_int, [P, P, P, P, P, P, P, P, P, c_int64, c_int64, c_void_p]),
    "nnhipFusedAdamWStep": (ctypes.c_int, [P, P, P, P, c_double, c_double, c_double, c_double, c_double, c_int32, c_int64, c_int32, c_float, c_void_p]),
    "nnhipCreateFusedOptimizer": (c_void_p, []),
    "nnhipDestroyFusedOptimizer": (ctypes.c_int, [c_void_p]),
    "nnhipFusedAdamWMultiTensorStep": (ctypes.c_int, [c_void_p, c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                                      POINTER(c_void_p), POINTER(c_int64), c_double, c_double, c_double, c_double, c_double,
                                                      c_int32, c_int32, c_float, c_void_p]),
    "nnhipFusedOptimizerSetStep": (ctypes.c_int, [c_void_p, c_int32, c_void_p]),
    "nnhipFusedOptimizerSetHyper": (ctypes.c_int, [c_void_p, c_double, c_double, c_float, c_void_p]),
    "nnhipFusedOptimizerSetGradDivisor": (ctypes.c_int, [c_void_p, P]),
    "nnhipConv2dForward": (ctypes.c_int, [P, P, P, P, POINTER(Conv2dDesc), c_void_p]),
    "nnhipConv2dBackward": (ctypes.c_int, [P, P, P, P, P, P, POINTER(Conv2dDesc), c_void_p]),
    "nnhipLeakyReLUForward": (ctypes.c_int, [P, P, c_float, c_int64, c_void_p]),
    "nnhipLeakyReLUBackward": (ctypes.c_int, [P, P, P, c_float, c_int64, c_void_p]),
    "nnhipSigmoidForward": (ctypes.c_int, [P, P, c_int64, c_void_p]),
    "nnhipSigmoidBackward": (ctypes.c_int, [P, P, P, c_int64, c_void_p]),
    "nnhipMaxPool2dForward": (ctypes.c_int, [P, P, P, POINTER(Pool2dDesc), c_void_p]),
    "nnhipMaxPool2dBackward": (ctypes.c_int, [P, P, P, POINTER(Pool2dDesc), c_void_p]),
    "nnhipMaxPool2dLeakyForward": (ctypes.c_int, [P, P, P, c_float, POINTER(Pool2dDesc), c_void_p]),
    "nnhipMaxPool2dLeakyBackward": (ctypes.c_int, [P, P, P, P, c_float, POINTER(Pool2dDesc), c_void_p]),
    "nnhipConv2dWeightGradPooledOk": (ctypes.c_int, [POINTER(Conv2dDesc), POINTER(Pool2dDesc)]),
    "nnhipConv2dWeightGradPooled": (ctypes.c_int, [P, P, P, P, c_float, P, P, POINTER(Conv2dDesc), POINTER(Pool2dDesc), c_void_p]),
    "nnhipConv2dLeakyMaxPoolForwardOk": (ctypes.c_int, [POINTER(Conv2dDesc), POINTER(Pool2dDesc)]),
    "nnhipConv2dLeakyMaxPoolForward": (ctypes.c_int, [P, P, P, c_float, P, P, POINTER(Conv2dDesc), POINTER(Pool2dDesc), c_void_p]),
    "nnhipConv2dLeakyMaxPoolStatsBlocks": (ctypes.c_int, [POINTER(Conv2dDesc), POINTER(Pool2dDesc)]),
    "nnhipConv2dLeakyMaxPoolForwardStats": (ctypes.c_int, [P, P, P, c_float, P, P, POINTER(Conv2dDesc), POINTER(Pool2dDesc), P, c_void_p]),
    "nnhipBatchNorm2dForward": (ctypes.c_int, [P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_float, c_float, ctypes.c_int, c_void_p]),
    "nnhipBatchNorm2dBackward": (ctypes.c_int, [P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipBatchNorm2dLinearSigmoidMSEFits": (ctypes.c_int, [c_int64, c_int64, c_int64, c_int64]),
    "nnhipBatchNorm2dLinearSigmoidMSE": (ctypes.c_int, [P, P, c_int64, c_int64, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, c_float,
                                                         c_float, P, P, c_int64, P, P, P, P, c_void_p]),
    "nnhipMSELossForwardBackward": (ctypes.c_int, [P, P, P, P, c_int64, c_void_p]),
    "nnhipMSELossSigmoidForwardBackward": (ctypes.c_int, [P, P, P, P, c_int64, c_void_p]),
    "nnhipScale": (ctypes.c_int, [P, c_float, c_int64, c_void_p]),
    "nnhipScaleRows": (ctypes.c_int, [P, P, P, c_int64, c_int64, c_int64, c_void_p]),
    "nnhipAdd": (ctypes.c_int, [P, P, P, c_int64, c_void_p]),
    "nnhipCommUniqueId": (ctypes.c_int, [ctypes.c_char_p]),
    "nnhipCommInitRank": (ctypes.c_int, [POINTER(c_void_p), ctypes.c_char_p, ctypes.c_int, ctypes.c_int]),
    "nnhipCommDestroy": (ctypes.c_int, [c_void_p]),
    "nnhipCommRank": (ctypes.c_int, [c_void_p, POINTER(ctypes.c_int), POINTER(ctypes.c_int)]),
    "nnhipCommLibrary": (ctypes.c_int, [ctypes.c_char_p, c_int64, POINTER(ctypes.c_int)]),
    "nnhipAllReduceSumF32": (ctypes.c_int, [c_void_p, P, c_int64, c_void_p]),
    "nnhipAllReduceAvgF32": (ctypes.c_int, [c_void_p, P, c_int64, c_void_p]),
    "nnhipBroadcastF32": (ctypes.c_int, [c_void_p, P, c_int64, ctypes.c_int, c_void_p]),
}
_NO_STATUS = {"nnhipVersion", "nnhipLinearReLULinearBackwardFits", "nnhipBatchNorm2dLinearSigmoidMSEFits", "nnhipConv2dLeakyMaxPoolStatsBlocks", "nnhipGetLastErrorString", "nnhipCreateFusedOptimizer", "nnhipGetGemmMode", "nnhipGetGemmLockstep", "nnhipConv2dWeightGradPooledOk", "nnhipConv2dLeakyMaxPoolForwardOk", "nnhipGemmLaunchCount",
              "nnhipWeightGradPending"}

_dll = None
_funcs: dict = {}


def lib_path() -> str:
    return os.environ.get(LIB_ENV, DEFAULT_LIB)


def load_library():
    """ctypes.CDLL(path, RTLD_GLOBAL) like utils.py:64-70 -- but a missing library is an error, not a print."""
    global _dll
    if _dll is None:
        path = lib_path()
        if not os.path.exists(path):
            raise NeunetHipError(
                f"libneunet_hip.so not found at {path!r}; build it with "
                f"`python numpy-nn-model_amd/build.py` (or set ${LIB_ENV}). There is no CPU fallback.")
        _dll = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    return _dll


def load_hip_function(name: str):
    """getattr + argtypes (utils.py:64-70)."""
    f = _funcs.get(name)
    if f is None:
        dll = load_library()
        try:
            f = getattr(dll, name)
        except AttributeError as exc:
            raise NeunetHipError(f"libneunet_hip.so does not export {name}") from exc
        restype, argtypes = _SIGNATURES[name]
        f.restype = restype
        f.argtypes = argtypes
        _funcs[name] = f
    return f


class StridedView:
    """A device array whose row stride travels as an explicit `ld` argument of the call (a column block of a wider
    buffer, e.g. q inside a fused q|k|v projection): to_pointer hands over its address without the contiguity check."""

    def __init__(self, tensor):
        if not tensor.is_cuda or tensor.stride(-1) != 1:
            raise ValueError("StridedView needs a device tensor with unit stride along its last dim")
        self.tensor = tensor


def to_pointer(obj):
    """Device array -> raw pointer (utils.py:72-82).  torch CUDA tensors play the role of CuPy arrays.
    NumPy arrays are rejected exactly like the reference does (utils.py:75-76)."""
    if obj is None:
        return None
    if isinstance(obj, StridedView):
        return obj.tensor.data_ptr()
    if hasattr(obj, "ctypes") and hasattr(obj, "__array_interface__"):
        raise TypeError("NumPy arrays are not supported here.")
    if hasattr(obj, "data_ptr"):
        if not obj.is_cuda:
            raise TypeError("Only device (cuda) tensors can be passed to HIP kernels.")
        if not obj.is_contiguous():
            raise ValueError("Device tensors passed to HIP kernels must be C-contiguous.")
        return obj.data_ptr()
    return obj


def get_current_stream_ptr():
    """Raw hipStream_t of torch's current stream (utils.py:87-92 took CuPy's current stream)."""
    import torch
    return torch.cuda.current_stream().cuda_stream


def last_error() -> str:
    s = load_hip_function("nnhipGetLastErrorString")()
    return s.decode() if s else ""


def call_hip_function(name: str, *args):
    """call_cuda_function (utils.py:84-85) + status check."""
    f = load_hip_function(name)
    rc = f(*[to_pointer(a) for a in args])
    if name not in _NO_STATUS and rc != 0:
        raise NeunetHipError(f"{name} failed with status {rc}: {last_error()}")
    if _wgrad["on"] and name in _WGRAD_ENTRIES:
        _wgrad_after_call(args)
    return rc


# ---- deferred parameter gradients (include/neunet_hip.h: nnhipWeightGradDefer) ---------------------------------------------------
# During Tensor.backward() the library queues the dW/db GEMMs of layers too small to fill the chip alone and launches them a
# two layers' worth at a time as ONE grid (GPT-tiny: four GEMMs per decoder layer, eight per flush).  The queued jobs read
# the X and dO buffers of the calls that queued them, so every array argument of those calls is kept referenced here until the
# flush -- otherwise torch's caching allocator could hand a dO buffer to the next kernel while a queued GEMM still has to read it.
_WGRAD_ENTRIES = {"nnhipLinearModuleBackward", "nnhipLinearModuleBackwardEx", "nnhipLinearModuleBackwardAct",
                  "nnhipLinearSwishBackward"}
_wgrad = {"on": False, "keep": [], "group": int(os.environ.get("NNHIP_WGRAD_GROUP", "8"))}


def _wgrad_after_call(args):
    pending = load_hip_function("nnhipWeightGradPending")()
    if pending == 0:
        _wgrad["keep"].clear()
        return
    _wgrad["keep"].append([a for a in args if hasattr(a, "data_ptr") or isinstance(a, StridedView)])
    if pending >= _wgrad["group"]:
        wgrad_flush(join=False)


def wgrad_begin() -> bool:
    """Start queueing (Tensor.backward); False if it is already on (nested backward) or switched off (NNHIP_WGRAD_GROUP < 2)."""
    if _wgrad["on"] or _wgrad["group"] < 2:
        return False
    call_hip_function("nnhipWeightGradDefer", 1, get_current_stream_ptr())
    _wgrad["on"] = True
    return True


# NNHIP_WGRAD_STREAM=1 (round 6, experimental): the grouped launch goes to a SIDE stream -- it depends on nothing the rest of the
# backward pass produces later and nothing before the optimizer reads it, so it can fill the launch gaps and ramps of the ~100
# dependent kernels of the main chain.  Fork: the side stream waits for an event recorded on the current stream (the queued calls'
# dO / X are complete); join: wgrad_end() (or a flush whose caller reads gradients next) makes the current stream wait for the side
# stream.  The queued calls' arrays stay referenced until the join.  Same kernels, same bits.
_WGRAD_SIDE = os.environ.get("NNHIP_WGRAD_STREAM", "0") == "1"
_side = {"stream": None, "busy": False, "keep": []}


def _side_stream():
    import torch
    if _side["stream"] is None:
        _side["stream"] = torch.cuda.Stream()
    return _side["stream"]


def wgrad_join():
    """Order the current stream behind the side stream's grouped launches (no-op when none is outstanding)."""
    if _side["busy"]:
        import torch
        done = torch.cuda.Event()
        done.record(_side["stream"])
        torch.cuda.current_stream().wait_event(done)
        _side["busy"] = False
        _side["keep"].clear()


def wgrad_flush(join: bool = True):
    """Launch what is queued (a layer's worth is there, a DP segment is about to be exchanged, or backward is over).
    join=False (the queue's own size trigger): with NNHIP_WGRAD_STREAM=1 the GEMMs may keep running on the side stream."""
    if _wgrad["on"]:
        if _WGRAD_SIDE and load_hip_function("nnhipWeightGradPending")() > 0:
            import torch
            side = _side_stream()
            ready = torch.cuda.Event()
            ready.record()
            side.wait_event(ready)
            call_hip_function("nnhipWeightGradFlushGemms", side.cuda_stream)
            _side["busy"] = True
            _side["keep"].extend(_wgrad["keep"])
        call_hip_function("nnhipWeightGradFlush", get_current_stream_ptr())
        _wgrad["keep"].clear()
    if join:
        wgrad_join()


def wgrad_end():
    _wgrad["on"] = False
    try:
        if _WGRAD_SIDE and load_hip_function("nnhipWeightGradPending")() > 0:
            _wgrad["on"] = True
            wgrad_flush(join=False)
            _wgrad["on"] = False
        call_hip_function("nnhipWeightGradDefer", 0, get_current_stream_ptr())
    finally:
        _wgrad["keep"].clear()
        wgrad_join()


def exported_symbols():
    return sorted(_SIGNATURES)
