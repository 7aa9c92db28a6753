#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kres.py rowops.hip [name-regex]
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(HERE, "numpy-nn-model_amd", "csrc")


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


def main():
    src = sys.argv[1]
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    extra = sys.argv[3:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-c", os.path.join(CSRC, src), "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage", *extra]
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    cur, rows = None, []
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key, rx in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"),
                        ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                        ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(rx, line)
            if m and cur is not None and key not in cur:
                cur[key] = int(m.group(1))
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-3000:])
        sys.exit(1)
    names = demangle([r_["name"] for r_ in rows])
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scr':>5} {'occ':>4} {'lds':>7}  kernel")
    for r_, n in zip(rows, names):
        n = n.replace("nnhip::", "").replace("void ", "")
        if pat and not pat.search(n):
            continue
        print(f"{r_.get('vgpr', -1):>5} {r_.get('agpr', -1):>5} {r_.get('sgpr', -1):>5} {r_.get('scratch', -1):>5} "
              f"{r_.get('occ', -1):>4} {r_.get('lds', -1):>7}  {n[:150]}")


if __name__ == "__main__":
    main()
