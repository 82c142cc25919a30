# kernel trace of the bench step -> per-step kernel table, queue timeline and the side-by-side lane dump of one step
# usage (GPU box): bash tools/trace_step.sh gpurun_out/<dir> [bench args]
set -e
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
O=$1; shift; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof.json 2>/dev/null
python tools/kstats.py $O/stats/st_kernel_trace.csv 80 > $O/kernel_time_per_step.txt
python tools/timeline.py $O/stats/st_kernel_trace.csv ${BACK:-12} > $O/timeline.txt
python tools/lane_dump.py $O/stats/st_kernel_trace.csv ${BACK:-12} > $O/lanes_one_step.txt
python tools/families.py $O/stats/st_kernel_trace.csv $O/kernel_families.json "$@" > $O/kernel_families.txt
cp $O/stats/st_kernel_stats.csv $O/rocprofv3_kernel_stats.csv 2>/dev/null || true
rm -rf $O/stats
head -3 $O/kernel_time_per_step.txt; cat $O/timeline.txt
