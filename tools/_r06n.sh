cd $GRAFT_REPO_ROOT
export NNHIP_ATTN_SB_FWD=pw
timeout 1500 bash tools/attn_sb_pmc.sh r06pw > gpurun_out/r06pw_attn_pmc.log 2>&1
tail -5 gpurun_out/r06pw_attn_pmc.log
ls gpurun_out/r06pw | head
