cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04e/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04e/gputest.log
tail -4 gpurun_out/r04e/gputest.log
python __graft_entry__.py smoke 2>&1 | tail -1
python tools/kbench.py --only convg > gpurun_out/r04e/kbench_convg.txt 2>&1; grep -v amdgpu gpurun_out/r04e/kbench_convg.txt | head -6
