#!/bin/bash
# developer tool: SQ counters of one GEMM kernel (tools/gemm_pmc_run.py), three --pmc passes.  usage: gemm_pmc.sh <tag> <mode> [M K N]
TAG=${1:-gemm}; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/gemm_pmc_$TAG; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d /tmp/gp1 -o a -- python $R/tools/gemm_pmc_run.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -f csv -d /tmp/gp2 -o b -- python $R/tools/gemm_pmc_run.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD -f csv -d /tmp/gp3 -o c -- python $R/tools/gemm_pmc_run.py "$@" > /dev/null 2>&1
for p in 1 2 3; do f=$(find /tmp/gp$p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/p$p.csv; done
python - <<PY
import csv, glob, collections
d = collections.defaultdict(list); dur = []
for f in sorted(glob.glob("$O/p*.csv")):
    for r in csv.DictReader(open(f)):
        if "gemm" not in r["Kernel_Name"]: continue
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
m = {k: sum(v) / len(v) for k, v in d.items()}
print("kernel us (with counters):", sum(dur) / max(len(dur), 1) / 1e3)
for k, v in sorted(m.items()): print(f"{k:28s} {v:16.0f}")
g = m.get("GRBM_GUI_ACTIVE", 0)
if g:
    simd = g / 8 * 1024
    print("MFMA busy %.1f %% of SIMD-cycles" % (100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd))
w = m.get("SQ_WAVE_CYCLES", 0)
if w:
    print("wave-cycles: issue-stall %.1f %%  parked %.1f %%  active %.1f %%" % (100 * m.get("SQ_WAIT_INST_ANY", 0) / w, 100 * m.get("SQ_WAIT_ANY", 0) / w, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / w))
mf = m.get("SQ_INSTS_MFMA", 0)
if mf:
    print("per MFMA: VALU %.2f  LDS %.2f  SALU %.2f ; LDS conflict/active %.3f" % ((m.get("SQ_INSTS_VALU", 0) - mf) / mf, m.get("SQ_INSTS_LDS", 0) / mf, m.get("SQ_INSTS_SALU", 0) / mf, m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
