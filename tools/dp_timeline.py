#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of a data-parallel step: which backend (RCCL) kernels ran, on which stream, and
which of our kernels were executing at the same time.

    python tools/dp_timeline.py <results.db | kernel_trace.csv> <out.md> [title]
"""
import collections
import csv
import sqlite3
import sys


def load(path):
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        return [dict(name=n, stream=s, start=a, end=b) for n, s, a, b in
                c.execute("select name, stream, start, end from kernels order by start")]
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append(dict(name=r["Kernel_Name"], stream=r.get("Stream_Id", r.get("Queue_Id", "?")),
                         start=int(r["Start_Timestamp"]), end=int(r["End_Timestamp"])))
    return sorted(rows, key=lambda r: r["start"])


def short(n, k=70):
    n = n.replace("void ", "").replace("nnhip::", "").replace("(anonymous namespace)::", "")
    return n if len(n) <= k else n[: k - 3] + "..."


def is_backend(n):
    n = n.lower()
    return "nccl" in n or "rccl" in n or "onerankreduce" in n or "msccl" in n


def main():
    path, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else "data-parallel step: backend kernels vs ours"
    rows = load(path)
    per_stream = collections.Counter(r["stream"] for r in rows)
    back = [r for r in rows if is_backend(r["name"])]
    ours = [r for r in rows if not is_backend(r["name"])]
    with open(out, "w") as f:
        f.write(f"# {title}\n\nsource: `{path}` ({len(rows)} kernel dispatches)\n\n## dispatches per stream\n\n| stream | dispatches | backend kernels |\n|---|---|---|\n")
        for s, n in per_stream.most_common():
            nb = sum(1 for r in back if r["stream"] == s)
            f.write(f"| {s} | {n} | {nb} |\n")
        names = collections.Counter(short(r["name"], 110) for r in back)
        f.write("\n## backend kernels\n\n| kernel | launches | avg us |\n|---|---|---|\n")
        for n, k in names.items():
            d = [r["end"] - r["start"] for r in back if short(r["name"], 110) == n]
            f.write(f"| `{n}` | {k} | {sum(d) / len(d) / 1e3:.1f} |\n")
        # the last complete step: walk back from the last backend kernel to the previous optimizer kernel
        if back:
            last = back[-1]
            opt = [r for r in ours if "adamw" in r["name"].lower()]
            prev_opt = [r for r in opt if r["end"] < back[max(0, len(back) - 1 - 4)]["start"]]
            t0 = prev_opt[-1]["end"] if prev_opt else rows[0]["start"]
            t1 = max([r["end"] for r in opt if r["start"] > last["start"]][:1] or [last["end"]])
            f.write(f"\n## last step ({(t1 - t0) / 1e6:.3f} ms from the previous optimizer kernel's end to this one's): every backend "
                    "kernel and what ran beside it\n\n| t (ms) | backend kernel (stream) | us | our kernels executing at the same time (stream) |\n|---|---|---|---|\n")
            for b in [r for r in back if t0 <= r["start"] <= t1]:
                con = [r for r in ours if r["start"] < b["end"] and r["end"] > b["start"]]
                txt = "; ".join(f"`{short(r['name'], 48)}` ({r['stream']})" for r in con[:4]) or "-- none: the stream was idle --"
                f.write(f"| {(b['start'] - t0) / 1e6:.3f} | `{short(b['name'], 40)}` ({b['stream']}) | {(b['end'] - b['start']) / 1e3:.1f} | {txt} |\n")
            n_step = sum(1 for r in ours if t0 <= r["start"] <= t1)
            f.write(f"\n{n_step} of our dispatches in that step; backend kernels overlapped by ours: "
                    f"{sum(1 for b in back if t0 <= b['start'] <= t1 and any(r['start'] < b['end'] and r['end'] > b['start'] for r in ours))}"
                    f" of {sum(1 for b in back if t0 <= b['start'] <= t1)}.\n")


if __name__ == "__main__":
    main()
