set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
O=${1:-gpurun_out/final}; mkdir -p $O    # MPMAE_COMMIT = commit under test (the GPU box has no .git)
python bench.py ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-400
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python bench.py ${BENCH_ARGS:-} --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmcf -o f --output-format csv -- python bench.py ${BENCH_ARGS:-} --mode eager --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmcw -o w --output-format csv -- python bench.py ${BENCH_ARGS:-} --mode eager --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmcf/f_counter_collection.csv $O/pmcw/w_counter_collection.csv $O/pmc_traffic.json > $O/pmc_traffic.txt
python tools/kstats.py $O/stats/st_kernel_trace.csv 70 > $O/kernel_time_per_step.txt
python tools/timeline.py $O/stats/st_kernel_trace.csv 3 > $O/timeline.txt
head -5 $O/kernel_time_per_step.txt; cat $O/timeline.txt | head -8
ls $O $O/stats
