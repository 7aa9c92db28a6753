#!/usr/bin/env python3
"""In-step kernel durations of the C4 GPT-tiny step, per GEMM / attention family, from a rocprofv3 kernel trace of
`python bench.py --workload c4` (the hipGraph replays: every dispatch between two adamw_multi_kernel launches is one step).

    python tools/c4_instep.py <..._kernel_trace.csv> profiles/c4_instep_families.json [collection tag]

bench.py's `also.c4_families` times every family ON ITS OWN with HIP events; inside the step the same kernels follow each other
without a gap, with other kernels' dirty lines in the L2s and the clocks of a chip that has been busy for seconds -- 4-7 % slower
(round-5 review).  This writes the in-step mean next to which bench.py prints its isolated figure.  A dispatch is assigned to a
family by kernel name, grid and -- where one kernel serves several shapes -- the nearest expected duration (the candidates differ
by >= 25 %)."""
import collections
import csv
import json
import math
import sys

# key -> (kernel-name substring, grid or None, expected us, launches per step, divisor: how many family units one launch covers)
FAMILIES = {
    "qkv_fwd": ("gemm_pst_kernel<0, true>", None, 190, 6, 1), "head_fwd": ("gemm_pst_kernel<0, true>", None, 1800, 1, 1),
    # (round 6: the forward saves swish'(z), EPI 3, and the input gradient multiplies by it, EPI 4; "|" = either kernel name)
    "fc1_fwd_swish": ("gemm_pst_kernel<3, true>|gemm_pst_kernel<1, true>", None, 260, 6, 1),
    "fc2_dx_swish": ("gemm_pst_kernel<4, false>|gemm_pst_kernel<2, false>", None, 255, 6, 1),
    "out_fwd": ("gemm_f32_kernel<32, true, true, true, false>", None, 72, 6, 1),
    "fc2_fwd": ("gemm_f32_kernel<32, true, true, true, false>", None, 250, 6, 1),
    "out_dx": ("gemm_f32_kernel<32, true, false, true, false>", None, 68, 6, 1),
    "qkv_dx": ("gemm_f32_kernel<32, true, false, true, false>", None, 184, 6, 1),
    "fc1_dx": ("gemm_f32_kernel<32, true, false, true, false>", None, 243, 6, 1),
    "head_dx": ("gemm_f32_kernel<32, true, false, true, false>", None, 1706, 1, 1),
    "layer_dw": ("gemm_f32_group_kernel", 393216, 1450, 3, 2), "head_dw": ("gemm_f32_group_kernel", 241664, 1817, 1, 1),
    "layer_dw_reduce": ("splitk_reduce_group_kernel", None, 23, 3, 2), "head_dw_reduce": ("splitk_reduce_kernel", None, 16, 1, 1),
    "attn_fwd": ("attn_sb_fwd_kernel", None, 52, 6, 1), "attn_bwd": ("attn_sb_bwd_kernel", None, 146, 6, 1),
    "rmsnorm_fwd": ("rmsnorm_fwd_rows", None, 12, 12, 1), "rmsnorm_bwd": ("rmsnorm_bwd_rows", None, 23, 12, 1),
    "ce": ("ce_rows_kernel", None, 345, 1, 1), "adam": ("adamw_multi_kernel", None, 155, 1, 1),
}


def main():
    path, out = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else "?"
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "adamw_multi_kernel" in r["Kernel_Name"]]
    # the eager discovery / warm-up passes launch torch copies and fills between our kernels; a graph replay does not: keep the steps
    # whose dispatch count equals the most common one
    steps = [rows[a + 1: b + 1] for a, b in zip(ends[:-1], ends[1:])]
    n_common = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if len(s) == n_common]
    per = collections.defaultdict(list)
    step_ms = []
    for s in steps:
        step_ms.append((int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"])) / 1e6)
        for r in s:
            us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            name, grid = r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))
            cands = [(abs(math.log(us / e)), k) for k, (sub, g, e, _, _) in FAMILIES.items()
                     if any(x in name for x in sub.split("|")) and (g is None or g == grid)]
            if cands:
                per[min(cands)[1]].append(us)
    res = {"_collected": tag, "_steps": len(steps), "_dispatches_per_step": n_common, "_step_ms_mean": round(sum(step_ms) / len(step_ms), 4),
           "_source": "rocprofv3 --kernel-trace of `python bench.py --workload c4` (tools/c4_instep.py): mean kernel duration inside the replayed step"}
    for k, (sub, g, e, n, div) in FAMILIES.items():
        v = per.get(k, [])
        if not v:
            continue
        if len(v) != n * len(steps):
            res.setdefault("_warnings", []).append(f"{k}: {len(v)} dispatches, expected {n} x {len(steps)}")
        res[k] = {"us": round(sum(v) / len(v) / div, 2), "launch_us": round(sum(v) / len(v), 2), "n": len(v), "min_us": round(min(v), 1), "max_us": round(max(v), 1),
                  "kernel": sub, "units_per_launch": div, "launches_per_step": n}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        print(k, v)


if __name__ == "__main__":
    main()
