#!/bin/bash
# Local post-processing of gpurun_out/<tag>/ (rocprofv3 CSVs) into the tracked profiles/ directory.  Usage: tools/make_profiles.sh r02
TAG=${1:-r02}; G=gpurun_out/$TAG; P=profiles; mkdir -p $P
for W in headline c1 c2 c3 c4 c5; do
  F=$(find $G/prof_$W -name "*_kernel_stats.csv" | head -1)
  [ -n "$F" ] && python tools/prof_summary.py stats $F $P/${TAG}_${W}_kernel_stats.md "Round ${TAG#r} -- rocprofv3 --kernel-trace --stats -- python bench.py --workload $W --no-cpu-baseline (MI355X)"
done
SQ=$(find $G/pmc_sq_c2 -name "*counter_collection.csv" | head -1)
[ -n "$SQ" ] && python tools/prof_summary.py pmc $SQ $P/${TAG}_c2_pmc_sq.md
for W in c2 c3; do
  FF=$(find $G/pmc_fetch_$W -name "*counter_collection.csv" | head -1); WW=$(find $G/pmc_write_$W -name "*counter_collection.csv" | head -1)
  python tools/prof_summary.py pmc $FF $WW $P/${TAG}_${W}_pmc_hbm.md
done
FF=$(find $G/pmc_fetch_c2 -name "*counter_collection.csv" | head -1); WW=$(find $G/pmc_write_c2 -name "*counter_collection.csv" | head -1)
python tools/prof_summary.py traffic $FF $WW /tmp/t_c2.json "gemm_f32_kernel<32, true, true, true, false>=gemm_fwd_c2" "gemm_f32_kernel<32, true, false, true, false>=gemm_dx_c2" "gemm_f32_kernel<32, false, false, true, false>=gemm_dw_c2" "adamw_multi=adamw_c2" > /dev/null
FF=$(find $G/pmc_fetch_c3 -name "*counter_collection.csv" | head -1); WW=$(find $G/pmc_write_c3 -name "*counter_collection.csv" | head -1)
python tools/prof_summary.py traffic $FF $WW /tmp/t_c3.json "map1_kernel<SwishF>%0=swish_fwd_c3" "map2_kernel<SwishB>%0=swish_bwd_c3" "rmsnorm_fwd_rows%0=rmsnorm_fwd_c3" "rmsnorm_bwd_rows%0=rmsnorm_bwd_c3" "softmax_fwd_rows=softmax_fwd_c3" "softmax_bwd_rows=softmax_bwd_c3" "ce_rows_kernel=ce_c3" "adamw_multi@524288=adamw_c3" "adamw_multi@1638400=adamw_200x512x1024_c3" > /dev/null
FF=$(find $G/pmc_fetch_c4 -name "*counter_collection.csv" 2>/dev/null | head -1); WW=$(find $G/pmc_write_c4 -name "*counter_collection.csv" 2>/dev/null | head -1)
echo '{}' > /tmp/t_c4.json
[ -n "$FF" ] && [ -n "$WW" ] && python tools/prof_summary.py traffic $FF $WW /tmp/t_c4.json "gemm_f32_group_kernel@393216=gemm_group_dw2_c4" "gemm_f32_group_kernel@241664=gemm_head_dw_c4" "attn_sb_fwd_kernel=attn_sb_fwd_c4" "attn_sb_bwd_kernel=attn_sb_bwd_c4" "ce_rows_kernel=ce_c4" > /dev/null
python - <<PY
import json
t = {}
for f in ['/tmp/t_c2.json', '/tmp/t_c3.json', '/tmp/t_c4.json']:
    t.update(json.load(open(f)))
out = {k: v['hbm_bytes_per_launch'] for k, v in t.items()}
out['_detail'] = t
out['_collected'] = '${TAG}'
out['_note'] = ("bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes over bench.py; FETCH_SIZE counts "
                "L2->fabric reads (Infinity-Cache hits included), doubled per the gfx950 half-count correction "
                "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated there. MI355X, round ${TAG}.")
json.dump(out, open('$P/pmc_traffic.json', 'w'), indent=1)
for k, v in out.items():
    if not k.startswith('_'): print(f"{k:18s} {v/1e6:9.1f} MB")
PY
for W in headline c1 c2 c3 c4 c5 nb c4_forced_nccl c4_forced_nccl_ingraph; do [ -f $G/bench_$W.json ] && cp $G/bench_$W.json $P/${TAG}_bench_$W.json; done
[ -f $G/dp_timeline.md ] && cp $G/dp_timeline.md $P/${TAG}_dp_forced_nccl_timeline.md
F=$(find $G/prof_dp -name "*_kernel_stats.csv" 2>/dev/null | head -1)
[ -n "$F" ] && python tools/prof_summary.py stats $F $P/${TAG}_c4_forced_nccl_kernel_stats.md "Round ${TAG#r} -- rocprofv3 --kernel-trace --stats -- python bench.py --workload c4 --force-dp --dp-op avg (1-rank nccl group; MI355X)"
# round 4: native-comm DP step, lock-step on/off (step time + fabric reads of the step's GEMM kernels), MFMA conv layer
for W in c4_native_comm c4_forced_lockstep0 c4_forced_lockstep1; do [ -f $G/bench_$W.json ] && cp $G/bench_$W.json $P/${TAG}_bench_$W.json; done
F=$(find $G/prof_conv -name "*_kernel_stats.csv" 2>/dev/null | head -1)
[ -n "$F" ] && python tools/prof_summary.py stats $F $P/${TAG}_conv_mfma_kernel_stats.md "Round ${TAG#r} -- rocprofv3 --kernel-trace --stats -- python tools/conv_prof.py 64 64 56 128 (Conv2d 64->128 ch, 56x56, batch 64: forward, dgrad, wgrad+db; 29.6 GFLOP each; MI355X)"
C=$(find $G/pmc_conv -name "*counter_collection.csv" 2>/dev/null | head -1)
[ -n "$C" ] && python tools/prof_summary.py pmc $C $P/${TAG}_conv_mfma_pmc_sq.md
python - <<PY
import csv, glob, json, collections
out = {}
for ls in (0, 1):
    fs = glob.glob("$G/pmc_fetch_c4_ls%d/**/*counter_collection.csv" % ls, recursive=True)
    if not fs: continue
    per = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != "FETCH_SIZE": continue
        k = r["Kernel_Name"].split("(")[0].replace("void nnhip::", "")
        if not k.startswith("gemm"): continue
        per[k] += float(r["Counter_Value"]); n[k] += 1
    steps = 3.0   # 1 warm-up + 2 timed steps + the discovery pass run eagerly: report per-launch means instead of per-step sums
    out["lockstep%d" % ls] = {k: {"launches": n[k], "fetch_MB_per_launch": round(2 * per[k] * 1024 / n[k] / 1e6, 1)} for k in sorted(per)}
    b = "$G/bench_c4_forced_lockstep%d.json" % ls
    try: out["lockstep%d" % ls]["_ms_per_step"] = json.load(open(b))["ms_per_step"]
    except Exception: pass
if out:
    out["_note"] = "C4 GPT-tiny step through the forced 1-rank nccl group (--force-dp --dp-op avg), NNHIP_GEMM_LOCKSTEP=0/1: L2->fabric reads per GEMM launch = 2*FETCH_SIZE KiB (gfx950 half-count correction), rocprofv3 --pmc FETCH_SIZE pass; ms_per_step from the un-profiled run"
    json.dump(out, open("$P/${TAG}_c4_lockstep_fabric.json", "w"), indent=1)
    for k, v in out.items():
        if not k.startswith("_"): print(k, v.get("_ms_per_step"), {kk: vv for kk, vv in v.items() if not kk.startswith("_")})
PY
for f in gemm_pmc_bf3.txt gemm_pmc_f32.txt; do [ -f $G/$f ] && cp $G/$f $P/${TAG}_$f; done
cp $G/kbench.log $P/${TAG}_kbench.txt
cp $G/stream_roof.txt $P/${TAG}_stream_roof.txt
for f in kbench_bf16x3.log gemm_accuracy.txt mfma_valu_probe.txt; do [ -f $G/$f ] && grep -v amdgpu.ids $G/$f > $P/${TAG}_${f%.*}.txt; done
# round 5: the balanced T = 256 attention kernels (attention_sb.hip): per-launch counter means (tools/attn_sb_pmc.sh)
[ -f $G/attn_pmc_summary.md ] && cp $G/attn_pmc_summary.md $P/${TAG}_attention_pmc.md
# round 6: in-step durations of the C4 families (bench.py prints them next to its isolated figures), the R/W stream probe, the n = 3 oracle step
T4=$(find $G/prof_c4 -name "*_kernel_trace.csv" 2>/dev/null | head -1)
[ -n "$T4" ] && python tools/c4_instep.py $T4 $P/c4_instep_families.json ${TAG} > /dev/null
[ -f $G/stream_nm_probe.txt ] && grep -v amdgpu.ids $G/stream_nm_probe.txt > $P/${TAG}_stream_nm_probe.txt
[ -s $G/cpu_c4_full_batch.json ] && python -c "import json,sys; json.load(open('$G/cpu_c4_full_batch.json'))" && cp $G/cpu_c4_full_batch.json $P/cpu_c4_full_batch.json
[ -f $G/gemm_pmc_pst.txt ] && cp $G/gemm_pmc_pst.txt $P/${TAG}_gemm_pmc_pst.txt
[ -f $G/attn_sb_check.log ] && tail -3 $G/attn_sb_check.log > $P/${TAG}_attn_sb_check.txt
