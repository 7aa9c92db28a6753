#!/usr/bin/env python3
"""Developer tool: where do the cycles of the fused attention kernels go?
Builds csrc/attention.hip a second time with -DAT_PROF (per-wave cycle counters around the kernel's phases) into
tools/attn_prof/libattn_prof.so (`--build`, needs hipcc; run it in the build container so the .so travels to the GPU box),
then runs the C4 shape (B64 T256 H8 dh64, pad+causal) and prints the mean cycles per phase, split by block weight and wave.
Phases (forward): 0 prologue, 1 wait at the tile-top barrier, 2 commit + barrier, 3 next-tile fetch issue, 4 S = K Q^T MFMAs,
5 mask + online softmax + rescale, 6 O += V^T P MFMAs, 7 epilogue."""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "attn_prof", "libattn_prof.so")
CSRC = os.path.join(ROOT, "numpy-nn-model_amd", "csrc")


def build():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-O3", "-DAT_PROF", "-shared",
           os.path.join(CSRC, "attention.hip"), os.path.join(CSRC, "runtime.hip"), "-o", SO]
    print(" ".join(cmd))
    subprocess.check_call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--kernel", default="fwd", choices=["fwd", "dkdv", "dq"])
    ap.add_argument("--shape", default="64,256,8,64")
    args = ap.parse_args()
    if args.build:
        build()
        return
    import numpy as np
    import torch
    lib = ctypes.CDLL(SO)
    B, T, H, dh = map(int, args.shape.split(","))
    D = H * dh
    dev = "cuda"
    torch.manual_seed(0)
    qkv = torch.randn(B, T, 3 * D, device=dev)
    dqkv = torch.empty_like(qkv)
    kvalid = torch.ones(B, T, dtype=torch.int32, device=dev)
    ctx, dctx = torch.empty(B, T, D, device=dev), torch.randn(B, T, D, device=dev)
    lse = torch.empty(B, H, T, 2, device=dev)
    NW = 4 if os.environ.get("NNHIP_ATTN_WAVES", "2") == "4" or dh != 64 else 2     # waves (32-row groups) per block
    nblk = ((B * H + 7) // 8) * 8 * ((T + 32 * NW - 1) // (32 * NW))
    prof = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device=dev)
    P = ctypes.c_void_p
    i64, i32, f32 = ctypes.c_int64, ctypes.c_int32, ctypes.c_float
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + 4 * off)  # noqa: E731
    sc = 1.0 / float(np.sqrt(D))
    lib.nnhipAttentionForward.argtypes = [P, P, P, P, P, P, i64, i64, i64, i64, i64, i64, f32, i32, P]
    lib.nnhipAttentionBackward.argtypes = [P, P, P, P, P, P, P, P, P, P, i64, i64, i64, i64, i64, i64, f32, i32, P]
    lib.nnhipAttentionSetProfile.argtypes = [P]

    def fwd():
        return lib.nnhipAttentionForward(ptr(qkv), ptr(qkv, D), ptr(qkv, 2 * D), ptr(kvalid), ptr(ctx), ptr(lse), B, H, T, T, dh,
                                         3 * D, sc, 1, st)

    def bwd():
        return lib.nnhipAttentionBackward(ptr(qkv), ptr(qkv, D), ptr(qkv, 2 * D), ptr(kvalid), ptr(ctx), ptr(dctx), ptr(lse),
                                          ptr(dqkv), ptr(dqkv, D), ptr(dqkv, 2 * D), B, H, T, T, dh, 3 * D, sc, 1, st)

    lib.nnhipAttentionSetProfile(None)
    for _ in range(5):
        assert fwd() == 0
        assert bwd() == 0
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    run = fwd if args.kernel == "fwd" else bwd
    a.record()
    for _ in range(20):
        run()
    e.record()
    torch.cuda.synchronize()
    print(f"{args.kernel}: {a.elapsed_time(e) / 20 * 1e3:.1f} us per call (instrumented build, profile buffer off)")
    lib.nnhipAttentionSetProfile(ptr(prof))
    os.environ["NNHIP_ATTN_PROF_KERNEL"] = args.kernel
    run()
    torch.cuda.synchronize()
    lib.nnhipAttentionSetProfile(None)
    pr = prof.cpu().numpy().reshape(nblk, 4, 8).astype(np.float64)
    nb = (T + 32 * NW - 1) // (32 * NW)
    nbx = (B * H + 7) // 8
    ids = np.arange(nblk)
    r = ids >> 3
    j = r // nbx                               # 0 = heaviest
    tune = 0
    if nb == 2 and (tune & 4):                 # map_block's mixed order
        A = nbx >> 1
        j = np.where(r < 2 * A, r & 1, np.where(r < 2 * A + (nbx - A), 0, 1))
    names = ["prologue", "barrier wait", "commit+barrier", "fetch issue", "MFMA phase 1", "VALU phase", "MFMA phase 2", "epilogue"]
    for lvl in range(nb):
        sel = pr[j == lvl]
        tot = sel[:, :NW].sum(axis=2)
        print(f"-- blocks of weight level {lvl} ({sel.shape[0]} blocks): mean total cycles per wave {tot.mean():.0f}")
        for w in range(NW):
            row = "  wave %d: " % w + "  ".join(f"{names[i]} {sel[:, w, i].mean():7.0f}" for i in range(8))
            print(row)


if __name__ == "__main__":
    main()
