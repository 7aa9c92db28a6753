cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_c4
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_c4 -o c4 -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 8 --warmup 5 --no-cpu-baseline > $O/prof_c4.log 2>&1
find $O -name "*.db" -delete
ls $O/prof_c4
