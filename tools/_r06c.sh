cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
timeout 600 bash tools/gemm_pmc.sh pst_qkv 0 16384 512 1536 > $O/pmc_pst_qkv.txt 2>&1; cat $O/pmc_pst_qkv.txt | tail -30
timeout 600 bash tools/gemm_pmc.sh pst_k512_n4096 0 16384 512 4096 > $O/pmc_pst_n4096.txt 2>&1; tail -6 $O/pmc_pst_n4096.txt
GEMM_PMC_OP=dx timeout 600 bash tools/gemm_pmc.sh f32_dx_512 0 16384 512 512 > $O/pmc_f32_dx512.txt 2>&1; tail -8 $O/pmc_f32_dx512.txt
