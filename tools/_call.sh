cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "conv" -x 2>&1 | tail -3
python tools/kbench.py --only convu 2>&1 | grep -v amdgpu > gpurun_out/r04c_kbench_convu.txt; cat gpurun_out/r04c_kbench_convu.txt | head -30
