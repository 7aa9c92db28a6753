#!/bin/bash
# developer tool: SQ counters for the fused attention kernels (tools/attn_abl.py workload); two --pmc passes
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/attn_pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d /tmp/ap1 -o a -- python $R/tools/attn_abl.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -f csv -d /tmp/ap2 -o b -- python $R/tools/attn_abl.py > /dev/null 2>&1
cp $(find /tmp/ap1 -name "*counter_collection.csv") $R/gpurun_out/attn_pmc/a_counter_collection.csv
cp $(find /tmp/ap2 -name "*counter_collection.csv") $R/gpurun_out/attn_pmc/b_counter_collection.csv
