# timeline of a TIMING EXPERIMENT (results invalid): usage bash tools/timeline_skip.sh <tag> "<skip substrings>"
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/tl_$1; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python tools/timing_experiment.py --skip "$2" -- --steps 30 --warmup 8 --no-cpu-baseline > /dev/null 2>&1
python tools/timeline.py $O/stats/st_kernel_trace.csv 10 > $O/timeline.txt
python tools/lane_dump.py $O/stats/st_kernel_trace.csv 10 > $O/lanes.txt
rm -rf $O/stats
head -8 $O/timeline.txt | cut -c1-170
