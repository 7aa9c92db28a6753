O=gpurun_out/r03h; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "^FAILED|^ERROR|passed|failed|AssertionError:"  $O/gpu_tests.log
