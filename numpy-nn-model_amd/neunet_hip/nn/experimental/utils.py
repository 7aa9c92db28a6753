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
