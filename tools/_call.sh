O=gpurun_out/r03m; mkdir -p $O
(time python bench.py --cpu-full-batch) > $O/cpu_full.json 2> $O/cpu_full.err; cat $O/cpu_full.json; tail -3 $O/cpu_full.err
