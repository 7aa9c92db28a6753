#!/bin/bash
# developer tool: L2 -> fabric read traffic (FETCH_SIZE) of one Linear GEMM per configuration, default vs lock-step mode.
# usage: gemm_fetch.sh <tag> "<lockstep 0|1> <fwd|dx|dw> <M> <K> <N>" ...
TAG=${1:-fetch}; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/gemm_fetch_$TAG; mkdir -p $O
i=0
for cfg in "$@"; do
  set -- $cfg; export NNHIP_GEMM_LOCKSTEP=$1; op=$2; M=$3; K=$4; N=$5; i=$((i+1))
  rm -rf /tmp/gf1
  # one counter per pass (FETCH_SIZE and WRITE_SIZE in one pass hung the profiler), every pass under its own timeout
  GEMM_PMC_OP=$op timeout 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d /tmp/gf1 -o a -- python $R/tools/gemm_pmc_run.py 0 $M $K $N > /dev/null 2>&1
  f=$(find /tmp/gf1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/c${i}.csv
  python - <<PY
import csv, collections
d = collections.defaultdict(list); dur = []
for r in csv.DictReader(open("$O/c${i}.csv")):
    if "gemm_f32_kernel" not in r["Kernel_Name"]: continue
    d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
m = {k: sum(v) / len(v) for k, v in d.items()}
M, K, N = $M, $K, $N
fetch = 2 * m.get("FETCH_SIZE", 0) * 1024 / 1e6          # gfx950: FETCH_SIZE counts half (MI355X_MICROARCH.md)
print("lockstep=$NNHIP_GEMM_LOCKSTEP $op M=%d K=%d N=%d  %d launches, %.1f us under the counter pass;  L2->fabric reads %.1f MB per launch  (operands %.1f MB, output %.1f MB)" % (
    M, K, N, len(dur), sum(dur) / max(len(dur), 1) / 1e3, fetch, 4 * (M * K + N * K) / 1e6, 4 * M * N / 1e6))
PY
done
