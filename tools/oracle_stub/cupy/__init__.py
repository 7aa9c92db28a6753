"""Stub `cupy` so the reference package (which does `import cupy` unconditionally,
neunet/autograd.py:3) can be imported on a box with no CUDA.  Only used by
tools/gen_golden.py in the build container; never shipped to / used on the GPU box."""


class ndarray:  # isinstance() target only
    pass


def __getattr__(name):
    raise AttributeError(f"stub cupy has no attribute {name!r} (CPU-only oracle import)")
