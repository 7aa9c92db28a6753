#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/...) into the tracked summaries under profiles/.

    python tools/prof_summary.py stats  <kernel_stats.csv> <out.md> [title]
    python tools/prof_summary.py pmc    <counter_collection.csv> [more.csv ...] <out.md>   (per-kernel counter means)
    python tools/prof_summary.py traffic <fetch_counter.csv> <write_counter.csv> <out.json> kernel_substr=tag ...
        HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024   -- FETCH_SIZE/WRITE_SIZE are in KiB and, on
        gfx950, FETCH_SIZE counts a wide coalesced stream at half its bytes (MI355X_MICROARCH.md, HBM section);
        WRITE_SIZE is uncalibrated there.  Collected in separate --pmc passes (TCC slots: 3 + 2 of 4).
"""
import collections
import csv
import json
import sys


def short(name, n=110):
    name = name.replace("void ", "").replace("nnhip::", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def stats(path, out, title="rocprofv3 --kernel-trace --stats"):
    rows = list(csv.DictReader(open(path)))
    with open(out, "w") as f:
        f.write(f"# {title}\n\nsource: `{path}`\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                    f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | "
                    f"{float(r['Percentage']):.2f} |\n")


def size_classes(durations, ratio=1.6):
    """Dispatches of one (kernel, grid) that differ in PROBLEM SIZE: rocprofv3 records no kernel arguments, and a persistent kernel
    uses one grid for every size (rmsnorm_bwd_rows: 131072 threads for the 8192-row and the 32768-row launches of `bench.py
    --workload c3` -- round 5 averaged the two into a 1.22 GB "traffic" figure for a 403 MB kernel).  Durations under profiling
    separate them: sorted, a jump by more than `ratio` starts a new class.  Returns {dispatch key: class index}, shortest first."""
    order = sorted(durations.items(), key=lambda kv: kv[1])
    cls, out, prev = 0, {}, None
    for key, us in order:
        if prev is not None and us > prev * ratio:
            cls += 1
        out[key] = cls
        prev = us
    return out


def load_counters(paths):
    """{(kernel, grid, size class): {counter: [values]}}, {same key: [durations us]} -- keyed on the problem size, not just the grid."""
    rows_by = collections.defaultdict(list)
    for p in paths:
        for r in csv.DictReader(open(p)):
            grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])
            rows_by[(short(r["Kernel_Name"], 90), grid)].append((p, r))
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for (name, grid), rows in rows_by.items():
        dd = {}
        for p, r in rows:
            dd[(p, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        cls = size_classes(dd)
        seen = set()
        for p, r in rows:
            key = (p, r["Dispatch_Id"])
            k = (name, grid, cls[key])
            d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if key not in seen:
                seen.add(key)
                dur[k].append(dd[key])
    return d, dur


def pmc(paths, out):
    d, dur = load_counters(paths)
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc per-kernel counter means\n\nsources: " + ", ".join(f"`{p}`" for p in paths) + "\n\n")
        for k in sorted(d, key=lambda k: -sum(dur[k])):
            if not k[0].startswith(("gemm", "map", "colsum", "softmax", "rmsnorm", "ce_", "adamw", "conv", "swiglu", "splitk")):
                continue
            ncls = 1 + max(kk[2] for kk in d if kk[:2] == k[:2])
            f.write(f"## `{k[0]}` grid={k[1]}" + (f" size class {k[2]} of {ncls} (by duration)" if ncls > 1 else "") +
                    f"  (n={len(dur[k])}, mean {sum(dur[k]) / len(dur[k]):.1f} us under profiling)\n\n")
            for c, v in sorted(d[k].items()):
                f.write(f"- {c}: {sum(v) / len(v):.6g}\n")
            v = d[k]
            if "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                cyc = sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"]) / 8
                mf = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(v["SQ_VALU_MFMA_BUSY_CYCLES"])
                f.write(f"- derived: MFMA pipe busy = {mf / 1024 / cyc * 100:.1f} % of cycles (1024 SIMDs; GRBM_GUI_ACTIVE/8 = {cyc:.4g} cycles), "
                        f"effective clock = {cyc / (sum(dur[k]) / len(dur[k])) / 1e3:.2f} GHz\n")
            f.write("\n")


def traffic(fetch_csv, write_csv, out, specs):
    fd, _ = load_counters([fetch_csv])
    wd, _ = load_counters([write_csv])
    res = {}
    for spec in specs:
        sub, tag = spec.split("=")
        grid, cls = None, None
        if "%" in sub:                      # kernel[@grid]%k: the k-th size class (by duration, shortest first) of that kernel / grid
            sub, cls = sub.split("%")
            cls = int(cls)
        if "@" in sub:
            sub, grid = sub.split("@")
            grid = int(grid)
        pick = lambda dd, c: [sum(v[c]) / len(v[c]) for k, v in sorted(dd.items(), key=lambda kv: kv[0][2])  # noqa: E731
                              if sub in k[0] and (grid is None or k[1] == grid) and (cls is None or k[2] == cls) and c in v]
        f, w = pick(fd, "FETCH_SIZE"), pick(wd, "WRITE_SIZE")
        if f and w:
            res[tag] = {"hbm_bytes_per_launch": (2 * f[0] + w[0]) * 1024, "FETCH_SIZE_KiB": f[0], "WRITE_SIZE_KiB": w[0],
                        "size_classes_seen": len(f), "selected": spec.split("=")[0],
                        "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE half-count correction)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "stats":
        stats(sys.argv[2], sys.argv[3], *(sys.argv[4:5]))
    elif mode == "pmc":
        pmc(sys.argv[2:-1], sys.argv[-1])
    elif mode == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:])
