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


def load_counters(paths):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = (short(r["Kernel_Name"], 90), int(r["Grid_Size"]))
            d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (p, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return d, dur


def pmc(paths, out):
    d, dur = load_counters(paths)
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc per-kernel counter means\n\nsources: " + ", ".join(f"`{p}`" for p in paths) + "\n\n")
        for k in sorted(d, key=lambda k: -sum(dur[k])):
            if not k[0].startswith(("gemm", "map", "colsum", "softmax", "rmsnorm", "ce_", "adamw", "conv", "swiglu", "splitk")):
                continue
            f.write(f"## `{k[0]}` grid={k[1]}  (n={len(dur[k])}, mean {sum(dur[k]) / len(dur[k]):.1f} us under profiling)\n\n")
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
        grid = None
        if "@" in sub:
            sub, grid = sub.split("@")
            grid = int(grid)
        pick = lambda dd, c: [sum(v[c]) / len(v[c]) for k, v in dd.items() if sub in k[0] and (grid is None or k[1] == grid) and c in v]  # noqa: E731
        f, w = pick(fd, "FETCH_SIZE"), pick(wd, "WRITE_SIZE")
        if f and w:
            res[tag] = {"hbm_bytes_per_launch": (2 * f[0] + w[0]) * 1024, "FETCH_SIZE_KiB": f[0], "WRITE_SIZE_KiB": w[0],
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
