#!/bin/bash
# developer tool: kernel durations + SQ / TA / TCP / TCC counters of the fused attention kernels (tools/attn_sb_run.py workload);
# one rocprofv3 --pmc pass per counter group (never combined with other trace domains; every pass under `timeout` -- a TA_* group
# hung a pass in round 5 and is left out).  Output: gpurun_out/<tag>/attn_pmc_*.csv
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --stats -f csv -d /tmp/ap0 -o s -- python $R/tools/attn_sb_run.py > /dev/null 2>&1
cp $(find /tmp/ap0 -name "*kernel_stats.csv") $OUT/attn_kernel_stats.csv
i=1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR"; do
  ATTN_REPS=12 timeout 150 rocprofv3 --kernel-trace --pmc $grp -f csv -d /tmp/ap$i -o p -- python $R/tools/attn_sb_run.py > /tmp/ap$i.log 2>&1
  f=$(find /tmp/ap$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp $f $OUT/attn_pmc_$i.csv; else echo "group $i failed: $grp"; tail -3 /tmp/ap$i.log; fi
  i=$((i+1))
done
python $R/tools/attn_sb_pmc_summary.py $TAG
