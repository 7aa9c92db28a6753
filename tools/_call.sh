cd $GRAFT_REPO_ROOT
timeout 2400 bash tools/collect_profiles.sh r04b > /dev/null 2>&1
ls gpurun_out/r04b | wc -l; du -sh gpurun_out/r04b
