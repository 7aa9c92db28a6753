#!/usr/bin/env python3
"""Build libneunet_hip.so (gfx950) in-tree with hipcc.

    python numpy-nn-model_amd/build.py [--force] [--debug]

hipcc cross-compiles without a GPU.  The .so lands in numpy-nn-model_amd/neunet_hip/lib/ (git-ignored,
but it travels with the repo snapshot to the GPU box).  Objects are cached under csrc/build/ and
rebuilt when a source or header is newer.
"""
import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "neunet_hip", "lib")
LIB = os.path.join(LIBDIR, "libneunet_hip.so")
SOURCES = ["runtime.hip", "gemm.hip", "gemm_small.hip", "gemm_bf3.hip", "gemm_pst.hip", "elementwise.hip", "rowops.hip", "optim.hip", "linear.hip", "conv2d.hip", "conv_mfma.hip", "embedding.hip", "pool_norm.hip", "attention.hip", "attention_sb.hip", "comm.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "adam_device.h"), os.path.join(CSRC, "gemm_small.h"), os.path.join(CSRC, "conv_common.h"), os.path.join(CSRC, "attention.h"),
           os.path.join(os.path.dirname(HERE), "include", "neunet_hip.h")]
ARCH = "gfx950"


# MFMA kernels whose inner loops must not touch scratch memory: a harmless-looking edit (an extra branch between the
# prologue loads and the first LDS store) once made the GEMM spill 144 B/lane and lose 30 % -- the build fails instead.
# (regexes on the mangled names; the GEMM's scalar-load variants -- VEC = false, unaligned operands -- are exempt.)
NO_SPILL = {"gemm.hip": (r"gemm_f32_kernelILi\d+ELb[01]ELb[01]ELb1E",),
            # (round 6: a branch around the flushing k-step INSIDE the k-loop spilled 172-432 B/lane; three loops in a row do not)
            "gemm_pst.hip": (r"gemm_pst_kernel",),
            # the tuned fast path: GEN = false (no dense mask / dropout); the general variants may spill a few registers
            "attention.hip": (r"attn_fwd_kernelILi\d+ELb0E", r"attn_bwd_dkdv_kernelILi\d+ELb0E", r"attn_bwd_dq_kernelILi\d+ELb0E"),
            "attention_sb.hip": (r"attn_sb_",)}
# The 2-wave-block attention kernels (head dim 64) sit exactly at the 256-VGPR limit of 2 waves per SIMD and keep two or
# three values in scratch (8-12 B/lane; measured 4-6 % FASTER than the 4-wave blocks all the same): tolerated up to here.
SPILL_ALLOWANCE = ((r"attn_\w+_kernelILi64ELb0ELi2E", 16),)


def check_no_spills(src, remarks):
    import re
    name, bad = None, []
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and any(re.search(k, name) for k in NO_SPILL[src]):
            allowed = max([a for pat, a in SPILL_ALLOWANCE if re.search(pat, name)] + [0])
            if int(m.group(1)) > allowed:
                bad.append((name, int(m.group(1))))
    # pass real diagnostics through (a remark is followed by its source line and a caret line: drop those too)
    lines, out, skip = remarks.splitlines(), [], 0
    for l in lines:
        if "remark:" in l:
            skip = 2
            continue
        if skip and (l.lstrip().startswith("|") or re.match(r"^\s*\d+ \|", l)):
            skip -= 1
            continue
        skip = 0
        if l.strip() and "remarks generated" not in l:
            out.append(l)
    if out:
        sys.stderr.write("\n".join(out) + "\n")
    if bad:
        raise RuntimeError(f"{src}: register spills in performance-critical kernels: {bad}")


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, debug=False, verbose=True, variant=None, defines=(), csrc=None):
    """variant / defines: an A/B build -- the same sources with extra -D flags into lib/libneunet_hip.<variant>.so (objects
    under csrc/build_<variant>/); select it at run time with NEUNET_HIP_LIB=<path>.  The default build is untouched."""
    CSRC = csrc or globals()["CSRC"]      # --csrc: another checkout's sources (A/B against an older commit)
    OBJ = os.path.join(globals()["CSRC"], "build" if not variant else f"build_{variant}")
    LIB = os.path.join(LIBDIR, "libneunet_hip.so" if not variant else f"libneunet_hip.{variant}.so")
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    cc = hipcc()
    flags = [f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-O3", "-Wall", "-Wno-unused-function"]
    flags += [f"-D{d}" for d in defines]
    if debug:
        flags += ["-g", "-save-temps=obj"]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        stale = force or newer(src, obj) or any(newer(h, obj) for h in HEADERS) or newer(__file__, obj)
        jobs.append((src, obj, stale))

    def compile_one(job):
        src, obj, stale = job
        if not stale:
            return obj
        cmd = [cc, *flags, "-c", src, "-o", obj]
        guard = os.path.basename(src) in NO_SPILL
        if guard:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        if verbose:
            print(" ".join(cmd), flush=True)
        if not guard:
            subprocess.run(cmd, check=True)
            return obj
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise subprocess.CalledProcessError(r.returncode, cmd)
        try:
            check_no_spills(os.path.basename(src), r.stderr)
        except Exception:
            os.remove(obj)      # so that the next build checks again
            raise
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        objs = list(ex.map(compile_one, jobs))
    if force or any(j[2] for j in jobs) or not os.path.exists(LIB):
        cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--debug", action="store_true")
    ap.add_argument("--variant", default=None, help="A/B build name: lib/libneunet_hip.<variant>.so")
    ap.add_argument("-D", dest="defines", action="append", default=[], help="extra preprocessor define (with --variant)")
    ap.add_argument("--csrc", default=None, help="with --variant: compile the sources of this directory instead of csrc/")
    a = ap.parse_args()
    print(build(a.force, a.debug, variant=a.variant, defines=a.defines, csrc=a.csrc))
