#!/bin/bash
# Run on the GPU box (via gpurun): benches + rocprofv3 kernel stats + PMC passes into gpurun_out/<tag>/.
# Post-process locally with tools/make_profiles.sh <tag> into profiles/.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python tools/kbench.py > $O/kbench.log 2>&1
timeout 600 python tools/stream_roof.py > $O/stream_roof.txt 2>&1
timeout 600 python tools/kbench.py --only gemm --gemm-mode 1 > $O/kbench_bf16x3.log 2>&1
timeout 600 python tools/gemm_accuracy.py > $O/gemm_accuracy.txt 2>&1
[ -x tools/probes/mfma_valu_probe ] && tools/probes/mfma_valu_probe > $O/mfma_valu_probe.txt 2>&1
# round 6: what a plain R-read / W-write stream reaches at the footprints of the HBM-bound kernels (cold and warm)
hipcc --offload-arch=gfx950 -O3 tools/probes/stream_nm_probe.hip -o /tmp/stream_nm_probe > /dev/null 2>&1 && timeout 300 /tmp/stream_nm_probe > $O/stream_nm_probe.txt 2>&1
# round 6: the hand-counted waits of attention_sb.hip against the fully drained build (-DSB_CHECK), and the wave-private-tile forward
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "counted_waits or wave_private" > $O/attn_sb_check.log 2>&1
# round 6: the n = 3 collection of the full-batch NumPy-oracle step that every bench line's live n = 1 sample cites
timeout 600 python bench.py --cpu-full-batch --cpu-full-steps 3 > $O/cpu_c4_full_batch.json 2> $O/cpu_c4_full_batch.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_headline.json 2> $O/bench_headline.err
timeout 600 python bench.py --workload c2 --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python bench.py --workload c3 --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err
timeout 600 python bench.py --workload c1 --steps 500 --warmup 50 > $O/bench_c1.json 2> $O/bench_c1.err
timeout 600 python bench.py --workload c4 --steps 10 --warmup 3 > $O/bench_c4.json 2> $O/bench_c4.err
timeout 600 python bench.py --workload c5 --steps 200 --warmup 20 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 600 python bench.py --workload nb --no-cpu-baseline > $O/bench_nb.json 2> $O/bench_nb.err
# the data-parallel step on the production backend with ONE rank: nccl (= RCCL) group, collectives forced
timeout 600 python bench.py --workload c4 --force-dp --no-cpu-baseline > $O/bench_c4_forced_nccl.json 2> $O/bench_c4_forced_nccl.err
timeout 600 python bench.py --workload c4 --force-dp --dp-ingraph 1 --no-cpu-baseline > $O/bench_c4_forced_nccl_ingraph.json 2> $O/bench_c4_forced_nccl_ingraph.err
cd /tmp && export TMPDIR=/tmp
for W in headline c1 c2 c3 c4 c5; do
  S=30; [ $W = c4 ] && S=8; [ $W = headline ] && S=8; [ $W = c1 ] && S=200; [ $W = c5 ] && S=100
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_$W -o $W -- python $R/bench.py --workload $W --steps $S --warmup 5 --no-cpu-baseline > $O/prof_$W.log 2>&1
done
for W in c2 c3; do
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_fetch_$W -o f -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_$W.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_write_$W -o w -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write_$W.log 2>&1
done
# the headline's dominant kernel (the grouped dW launch of two decoder layers) inside the C4 step: fabric reads and writes per launch
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_fetch_c4 -o f -- python $R/bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_c4.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_write_c4 -o w -- python $R/bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_c4.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $O/pmc_sq_c2 -o sq -- python $R/bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_sq_c2.log 2>&1
# RCCL's kernels next to ours: kernel trace of the forced one-rank nccl step (--dp-op avg: a 1-rank in-place SUM is elided inside RCCL)
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_dp -o dp -- python $R/bench.py --workload c4 --force-dp --dp-op avg --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_dp.log 2>&1
# the same DP step with the exchange through the library's own RCCL entry points (nnhipAllReduce*F32, NativeComm), and the
# lock-step GEMM default of N > 1 jobs on / off (step time; fabric bytes of the step's GEMMs in two FETCH_SIZE passes)
timeout 600 python $R/bench.py --workload c4 --force-dp --comm native --dp-op avg --no-cpu-baseline > $O/bench_c4_native_comm.json 2> $O/bench_c4_native_comm.err
for LS in 0 1; do
  NNHIP_GEMM_LOCKSTEP=$LS timeout 600 python $R/bench.py --workload c4 --force-dp --dp-op avg --no-cpu-baseline > $O/bench_c4_forced_lockstep$LS.json 2> $O/bench_c4_forced_lockstep$LS.err
  NNHIP_GEMM_LOCKSTEP=$LS timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_fetch_c4_ls$LS -o f -- python $R/bench.py --workload c4 --force-dp --dp-op avg --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_c4_ls$LS.log 2>&1
done
# MFMA-bound conv layers (conv_mfma.hip): kernel stats + SQ counters of one 64->128 ch 56x56 B64 layer
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_conv -o conv -- python $R/tools/conv_prof.py 64 64 56 128 20 > $O/prof_conv.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA -f csv -d $O/pmc_conv -o c -- python $R/tools/conv_prof.py 64 64 56 128 5 > $O/pmc_conv.log 2>&1
cd $R
timeout 1500 bash tools/attn_sb_pmc.sh $TAG > $O/attn_pmc.log 2>&1      # -> $O/attn_pmc_summary.md (balanced T=256 kernels of attention_sb.hip)
timeout 900 bash tools/gemm_pmc.sh ${TAG}_bf3 1 8192 4096 4096 > $O/gemm_pmc_bf3.txt 2>&1
timeout 900 bash tools/gemm_pmc.sh ${TAG}_f32 0 8192 4096 4096 > $O/gemm_pmc_f32.txt 2>&1
timeout 900 bash tools/gemm_pmc.sh ${TAG}_pst 0 16384 512 1536 > $O/gemm_pmc_pst.txt 2>&1      # the persistent K = 512 kernel (q|k|v forward of C4)
cd /tmp
# keep the merge under the 64 MiB limit: drop the raw traces, keep stats + counter CSVs
timeout 600 python $R/tools/dp_timeline.py $(find $O/prof_dp -name "*_kernel_trace.csv" | head -1) $O/dp_timeline.md "C4 step, 1-rank nccl (RCCL) process group, collectives forced, --dp-op avg: RCCL kernels vs ours (rocprofv3 --kernel-trace)" > /dev/null 2>&1
find $O -name "*_kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
find $O -name "*.db" -delete
du -sh $O
