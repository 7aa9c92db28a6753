#!/usr/bin/env python3
"""Developer tool: per-block cycle stamps of gemm_f32_kernel (prologue / k-loop / epilogue incl. store acknowledgement).
`--build` compiles csrc/gemm*.hip + runtime.hip with -DGEMM_PROF into tools/gemm_prof/libgemm_prof.so (in the build container);
without it the tool runs C = A B^T at the given shapes and prints, per generation of 512 blocks, when blocks start and how
long each phase takes (cycles of s_memtime ~ core clock)."""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "gemm_prof", "libgemm_prof.so")
CSRC = os.path.join(ROOT, "numpy-nn-model_amd", "csrc")


def build():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-O3", "-DGEMM_PROF", "-shared"] + \
          sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip")) + ["-o", SO]
    print(" ".join(cmd))
    subprocess.check_call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--placement", action="store_true")
    ap.add_argument("--shapes", default="16384x2048x512,16384x512x512,16384x512x2048,4096x4096x4096")
    args = ap.parse_args()
    if args.build:
        build()
        return
    import numpy as np
    import torch
    lib = ctypes.CDLL(SO)
    P, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
    lib.nnhipGemmF32Ex.argtypes = [P, P, P, P, i64, i64, i64, i64, i64, i64, i32, i32, i64, i64, i64, i64, i64, i64, i64, i64, f32, P]
    lib.nnhipGemmSetProfile.argtypes = [P]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for shp in args.shapes.split(","):
        M, N, K = map(int, shp.split("x"))
        A = torch.rand(M, K, device="cuda") - 0.5
        B = torch.rand(N, K, device="cuda") - 0.5
        C = torch.empty(M, N, device="cuda")
        nblk = ((M + 127) // 128) * ((N + 127) // 128)
        prof = torch.zeros(nblk * 4 * 12, dtype=torch.int64, device="cuda")

        def run():
            rc = lib.nnhipGemmF32Ex(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, K, K, N, 1, 1, 1, 0, 0, 0, 1, 0, 0, 0, 1.0, st)
            assert rc == 0, rc

        lib.nnhipGemmSetProfile(None)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run()
        e.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(e) / 20 * 1e3
        lib.nnhipGemmSetProfile(ctypes.c_void_p(prof.data_ptr()))
        run()
        torch.cuda.synchronize()
        lib.nnhipGemmSetProfile(None)
        raw = prof.cpu().numpy().reshape(nblk, 4, 12)
        t = raw.astype(np.float64)[:, 0, :]      # wave 0 of each block
        t0 = t[:, 0].min()
        start, pro, loop, epi, ack = t[:, 0] - t0, t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
        end = t[:, 4] - t0
        tot = (t[:, 4] - t[:, 0]).mean()
        e = t[:, 5:10]
        print("   epilogue (mean ticks): loop end -> transposition starts %.0f | group 0: LDS writes+wait %.0f, reads+stores %.0f | group 1: %.0f, %.0f | -> end %.0f"
              % ((e[:, 0] - t[:, 2]).mean(), (e[:, 1] - e[:, 0]).mean(), (e[:, 2] - e[:, 1]).mean(), (e[:, 3] - e[:, 2]).mean(),
                 (e[:, 4] - e[:, 3]).mean(), (t[:, 3] - e[:, 4]).mean()))
        print(f"== {M}x{N}x{K}: {us:.1f} us per launch ({2.0 * M * N * K / us / 1e6:.1f} TFLOP/s), {nblk} blocks "
              "(s_memtime stamps; the counters of different XCDs are not synchronised, only differences within a block mean something)")
        order = np.argsort(start)
        for g in range(0, nblk, 512):
            sel = order[g:g + 512]
            print(f"   blocks {g:5d}-{g + len(sel) - 1:5d} (by start stamp): prologue {pro[sel].mean():6.0f}  k-loop {loop[sel].mean():7.0f}  "
                  f"epilogue issue {epi[sel].mean():6.0f}  store ack {ack[sel].mean():6.0f}   (ticks; block lifetime {tot:.0f})")
            if g >= 512 * 5:
                print("   ...")
                break
        if args.placement:
            placement(raw, t)


def placement(raw, t):
    """where each block ran (HW_ID / XCC_ID) and how long the k-loops of the two blocks sharing a CU take (the stamps of different
    CUs are not comparable)"""
    import numpy as np
    hw = raw[:, 0, 10]
    xcc = (raw[:, 0, 11] >> 32) & 0xF
    cu, sh, se, wid = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7, hw & 0xF
    loop = t[:, 2] - t[:, 1]
    ids = np.arange(len(loop)) >> 3                   # dispatch index inside the XCD (block b runs on XCD b % 8)
    print("   XCC_ID == block mod 8 for %d of %d blocks" % (int((xcc == (np.arange(len(loop)) & 7)).sum()), len(loop)))
    for lo in range(0, int(ids.max()) + 1, 32):
        sel = (ids >= lo) & (ids < lo + 32)
        print(f"   dispatch index {lo:3d}-{lo + 31:3d} of each XCD: k-loop ticks mean {loop[sel].mean():9.0f}  min {loop[sel].min():9.0f}  max {loop[sel].max():9.0f}"
              f"   wave_id parity {np.bincount((wid[sel] & 1).astype(int), minlength=2)}")
    slot = xcc * 1024 + se * 32 + sh * 16 + cu
    per = {}
    for b in range(len(loop)):
        per.setdefault(int(slot[b]), []).append(b)
    rows = [sorted(v, key=lambda b: t[b, 0]) for v in per.values() if len(v) == 4]
    if rows:
        r = np.array(rows)
        base = t[r[:, 0], 0][:, None]
        print(f"   timeline of the {len(rows)} CUs that ran exactly four blocks (ticks after the CU's first block started; mean over CUs):")
        for i in range(4):
            b = r[:, i]
            print(f"      block {i + 1}: start {np.mean(t[b, 0] - base[:, 0]):9.0f}  k-loop {np.mean(t[b, 1] - base[:, 0]):9.0f} .. {np.mean(t[b, 2] - base[:, 0]):9.0f}"
                  f"  (length {np.mean(t[b, 2] - t[b, 1]):8.0f})  stores acknowledged {np.mean(t[b, 4] - base[:, 0]):9.0f}")
    g1 = ids < 64
    mates = {}
    for b in np.nonzero(g1)[0]:
        mates.setdefault(int(slot[b]), []).append(int(ids[b]))
    d = [abs(v[0] - v[1]) for v in mates.values() if len(v) == 2]
    print(f"   first 64 per XCD: {len(mates)} distinct CUs, {len(d)} hold exactly two blocks; dispatch-index distance of the two: {np.bincount(d)[np.unique(d)]} at {np.unique(d)}")


if __name__ == "__main__":
    main()
