cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04a/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04a/gputest.log
tail -40 gpurun_out/r04a/gputest.log
