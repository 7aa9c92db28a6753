#!/usr/bin/env python3
"""Where does a kernel's VGPR high-water mark come from?  Per basic block of the ISA: highest VGPR index touched.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only x.hip -o x.s ;  python tools/vgpr_map.py x.s <mangled-substr> [threshold]
"""
import re
import sys

s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 0
start = next(i for i, l in enumerate(s) if l.startswith("_ZN") and key in l and l.rstrip().endswith(tuple(": ;")) or (l.startswith("_ZN") and key in l and ":" in l))
end = next(i for i in range(start, len(s)) if ".end_amdhsa_kernel" in s[i] or s[i].startswith(".Lfunc_end"))
blk, cur, first, n = "entry", 0, start, 0
for i in range(start, end):
    l = s[i]
    if re.match(r"^\.LBB\d+_\d+:", l):
        if cur >= thr:
            print(f"{blk:14s} lines {first - start:5d}-{i - start:5d}  max v{cur}  ({n} instr)")
        blk, cur, first, n = l.split(":")[0], 0, i, 0
        continue
    if l.startswith("\t") and not l.startswith("\t."):
        n += 1
    regs = [int(x) for x in re.findall(r"\bv(\d+)\b", l)] + [int(b) for a, b in re.findall(r"v\[(\d+):(\d+)\]", l)]
    cur = max([cur] + regs)
print(f"{blk:14s} lines {first - start:5d}-{end - start:5d}  max v{cur}  ({n} instr)")
