set -x
O=gpurun_out/r03b; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q > $O/dp_tests.log 2>&1; echo "dp tests rc=$?"
tail -30 $O/dp_tests.log
timeout 600 python bench.py --workload c4 --no-cpu-baseline > $O/c4_plain.json 2> $O/c4_plain.err; echo rc=$?
timeout 600 python bench.py --workload c4 --force-dp --no-cpu-baseline > $O/c4_forcedp.json 2> $O/c4_forcedp.err; echo rc=$?
timeout 600 python bench.py --workload c4 --force-dp --dp-ingraph 1 --no-cpu-baseline > $O/c4_forcedp_ingraph.json 2> $O/c4_forcedp_ingraph.err; echo rc=$?
cat $O/c4_plain.json $O/c4_forcedp.json $O/c4_forcedp_ingraph.json | cut -c1-400
tail -n 5 $O/c4_forcedp.err; tail -n 5 $O/c4_forcedp_ingraph.err
timeout 600 python bench.py --workload c4 --force-dp --dp-op avg --no-cpu-baseline > $O/c4_forcedp_avg.json 2> $O/c4_forcedp_avg.err; echo rc=$?; cut -c1-300 $O/c4_forcedp_avg.json
R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_dp -o dp -- python $R/bench.py --workload c4 --force-dp --dp-op avg --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/prof_dp.log 2>&1; echo rc=$?
ls -R $R/$O/prof_dp | head
