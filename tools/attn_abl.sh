#!/bin/bash
# developer tool: per-kernel durations of tools/attn_abl.py under rocprofv3 (the argument list is just a set of run tags;
# it was used with in-kernel ablation switches while tuning)
cd /tmp && export TMPDIR=/tmp
for a in "$@"; do
  rm -rf /tmp/p_$a
  NNHIP_ATTN_ABL=$a rocprofv3 --kernel-trace --stats -f csv -d /tmp/p_$a -- python $GRAFT_REPO_ROOT/tools/attn_abl.py > /dev/null 2>&1
  f=$(find /tmp/p_$a -name "*kernel_stats.csv" | head -1)
  echo "ABL $a: $(grep attn $f | cut -d, -f1-6 | tr '\n' ' ')"
done
