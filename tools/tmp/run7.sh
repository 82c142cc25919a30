cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/tr1; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof.json 2>$O/err.txt
f=$O/stats/st_kernel_trace.csv
python tools/lane_dump.py $f 2 0 700 > $O/front.txt
python tools/lane_dump.py $f 2 1400 2300 > $O/mid.txt
python tools/timeline.py $f > $O/timeline.txt
python tools/kstats.py $f 70 > $O/kernel_time_per_step.txt
rm -rf $O/stats
