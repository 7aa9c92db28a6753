#!/usr/bin/env python3
"""gpurun_out/attn_pmc/{a,b}_counter_collection.csv (tools/attn_pmc.sh) -> profiles/<tag>_attention_pmc.md"""
import csv
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
vals = defaultdict(lambda: defaultdict(list))
for f in ("gpurun_out/attn_pmc/a_counter_collection.csv", "gpurun_out/attn_pmc/b_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn_" not in n:
            continue
        k = n.split("(")[0].replace("void nnhip::", "")
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        vals[k]["_dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        vals[k]["_vgpr"] = [float(r["VGPR_Count"])]
out = [f"# Round {tag[1:]} -- fused attention kernels, SQ counters (rocprofv3 --pmc, two separate passes over tools/attn_abl.py)", "",
       "Workload: B64 T256 H8 head_dim 64, causal, forward + backward, 40 repetitions; values are per-launch means.",
       "`MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs): the fraction of the kernel's SIMD-cycles the",
       "matrix pipe was occupied (every `v_mfma_f32_32x32x2_f32` holds it 64 cycles).  On gfx950 the fp32 MFMA runs on the vector ALU's",
       "lanes (DESIGN.md 5): the VALU instructions of the softmax are paid in the same SIMD-cycles, so `MFMA busy + VALU share` is the",
       "figure to read against 100 %, not MFMA busy alone.  Durations include the counter-collection overhead.", "",
       "| kernel | VGPRs | us (with counters) | MFMA insts | MFMA busy | VALU insts / MFMA | ~VALU share (4 cyc each) | LDS insts / MFMA | LDS bank-conflict / LDS active | wave: issue-stall | wave: parked |",
       "|---|---|---|---|---|---|---|---|---|---|---|"]
m = lambda d, k: sum(d[k]) / len(d[k]) if d.get(k) else float("nan")  # noqa: E731
for k, d in sorted(vals.items()):
    gui = m(d, "GRBM_GUI_ACTIVE")
    simd_cycles = gui / 8 * 1024
    busy = m(d, "SQ_VALU_MFMA_BUSY_CYCLES") / simd_cycles
    mf, va, lds = m(d, "SQ_INSTS_MFMA"), m(d, "SQ_INSTS_VALU"), m(d, "SQ_INSTS_LDS")
    valu_share = (va - mf) * 4 / 64 / simd_cycles * 64 if False else ((va - mf) * 4) / (simd_cycles * 1.0) / 1.0
    # SQ_INSTS_* count wave-instructions over the whole chip; one VALU wave-instruction occupies its SIMD ~4 cycles
    wave = m(d, "SQ_WAVE_CYCLES")
    out.append(f"| `{k}` | {int(d['_vgpr'][0])} | {m(d, '_dur') / 1e3:.1f} | {mf:.0f} | {busy * 100:.0f} % | {(va - mf) / mf:.1f} | {valu_share * 100:.0f} % | "
               f"{lds / mf:.2f} | {m(d, 'SQ_LDS_BANK_CONFLICT') / max(m(d, 'SQ_LDS_IDX_ACTIVE'), 1) * 100:.1f} % | "
               f"{m(d, 'SQ_WAIT_INST_ANY') / wave * 100:.0f} % | {(1 - m(d, 'SQ_ACTIVE_INST_VALU') * 4 / wave - m(d, 'SQ_WAIT_INST_ANY') / wave) * 100:.0f} % |")
open(f"profiles/{tag}_attention_pmc.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
