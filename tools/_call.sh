cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04d/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04d/gputest.log
tail -6 gpurun_out/r04d/gputest.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04d/bench_headline.json 2> gpurun_out/r04d/bench_headline.err ) 2>&1 | grep real
python __graft_entry__.py smoke 2>&1 | tail -2
