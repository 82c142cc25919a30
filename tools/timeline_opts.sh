cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/tl_$1; mkdir -p $O
MPMAE_ENGINE_OPTS="$2" timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
python tools/timeline.py $O/stats/st_kernel_trace.csv 6 > $O/timeline.txt
python tools/lane_dump.py $O/stats/st_kernel_trace.csv 6 > $O/lanes.txt
rm -rf $O/stats
head -9 $O/timeline.txt | cut -c1-170
