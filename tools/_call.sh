cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "conv" -x > gpurun_out/r04b/conv_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04b/conv_tests.log
tail -4 gpurun_out/r04b/conv_tests.log
timeout 300 python tools/kbench.py --only convg > gpurun_out/r04b/kbench_convg.log 2>&1
grep -v amdgpu.ids gpurun_out/r04b/kbench_convg.log | grep wgrad
