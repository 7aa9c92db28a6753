"""Developer tool: print the last N kernel dispatches (start offset us, duration us, name) of a rocprofv3 kernel trace.
    python tools/timeline.py <kernel_trace.csv> [N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -60:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1000:9.1f} {(e - s) / 1000:7.1f}  {r['Kernel_Name'][:80]}")
