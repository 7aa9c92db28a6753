#!/usr/bin/env python3
"""What a float4 stream reaches on THIS box when nothing can come out of the 256 MiB Infinity Cache: the practical HBM
roof that the 8 TB/s spec fractions in bench.py should be read against.  Buffers of 2 GiB (8x the MALL); the library's
own elementwise kernels (ReLU = 8 B/elem, add = 12 B/elem) and a torch copy.   python tools/stream_roof.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "numpy-nn-model_amd"))
import torch  # noqa: E402
from neunet_hip._lib import call_hip_function as call, get_current_stream_ptr  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    st = get_current_stream_ptr()
    for mib in (128, 512, 2048):
        n = mib * (1 << 20) // 4
        x = torch.randn(n, device="cuda")
        y = torch.empty_like(x)
        z = torch.randn(n, device="cuda")
        t = timeit(lambda: call("nnhipReLUForward", y, x, n, st))
        print(f"relu  fwd  {mib:5d} MiB/buffer  8 B/elem: {8 * n / t / 1e9:8.1f} GB/s ({8 * n / t / 8e12 * 100:5.1f}% of 8 TB/s)")
        t = timeit(lambda: call("nnhipAdd", y, x, z, n, st))
        print(f"add        {mib:5d} MiB/buffer 12 B/elem: {12 * n / t / 1e9:8.1f} GB/s ({12 * n / t / 8e12 * 100:5.1f}% of 8 TB/s)")
        t = timeit(lambda: y.copy_(x))
        print(f"torch copy {mib:5d} MiB/buffer  8 B/elem: {8 * n / t / 1e9:8.1f} GB/s ({8 * n / t / 8e12 * 100:5.1f}% of 8 TB/s)")
        del x, y, z


if __name__ == "__main__":
    main()
