bash tools/gemm_pmc.sh bf3 1 8192 4096 4096 2>&1 | tail -40
echo ======= fp32
bash tools/gemm_pmc.sh f32 0 8192 4096 4096 2>&1 | tail -40
