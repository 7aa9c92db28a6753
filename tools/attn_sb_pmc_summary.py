#!/usr/bin/env python3
"""gpurun_out/<tag>/attn_pmc_*.csv + attn_kernel_stats.csv (tools/attn_sb_pmc.sh) -> per-kernel means of every counter, as markdown on stdout
and gpurun_out/<tag>/attn_pmc_summary.md"""
import csv
import glob
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r05a"
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = os.path.join(root, "gpurun_out", tag)
vals = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "attn_pmc_*.csv"))):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn_" not in n:
            continue
        k = n.split("(")[0].replace("void nnhip::", "")
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        vals[k]["_dur_us(pmc)"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        vals[k]["_vgpr"] = [float(r["VGPR_Count"])]
        vals[k]["_lds"] = [float(r["LDS_Block_Size"])] if "LDS_Block_Size" in r else [0.0]
out = [f"# {tag} -- fused attention kernels, rocprofv3 counters (per-launch means; B64 T256 H8 dh64 causal)", ""]
st = os.path.join(d, "attn_kernel_stats.csv")
if os.path.exists(st):
    out += ["| kernel | calls | avg us (no counters) | min | max |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(st)):
        if "attn_" in r["Name"]:
            out.append(f"| `{r['Name'].split('(')[0].replace('void nnhip::', '')}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} |")
    out.append("")
m = lambda v: sum(v) / len(v)  # noqa: E731
for k, c in sorted(vals.items()):
    out.append(f"## `{k}`")
    gui = m(c["GRBM_GUI_ACTIVE"]) if "GRBM_GUI_ACTIVE" in c else float("nan")
    out.append("| counter | mean | note |")
    out.append("|---|---|---|")
    for name in sorted(c):
        v = m(c[name])
        note = ""
        if name == "SQ_VALU_MFMA_BUSY_CYCLES":
            note = f"MFMA busy = {v / (gui / 8 * 1024) * 100:.0f} % of SIMD-cycles"
        if name in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY") and "SQ_WAVE_CYCLES" in c:
            note = f"{v / m(c['SQ_WAVE_CYCLES']) * 100:.0f} % of wave-cycles"
        if name.startswith("SQ_INSTS_") and "SQ_INSTS_MFMA" in c:
            note = f"{v / m(c['SQ_INSTS_MFMA']):.2f} per MFMA"
        out.append(f"| {name} | {v:.4g} | {note} |")
    out.append("")
open(os.path.join(d, "attn_pmc_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
