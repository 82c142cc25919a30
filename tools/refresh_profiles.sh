# Round profile set of the headline workload (GPU box): bench line, rocprofv3 kernel stats + per-step table + queue timeline + lane dump +
# in-step kernel families, PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes, eager driver), piece / prefix timings.
# usage: MPMAE_COMMIT=<sha> [BENCH_ARGS="..."] bash tools/refresh_profiles.sh gpurun_out/<dir>
set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
O=${1:-gpurun_out/final}; mkdir -p $O    # MPMAE_COMMIT = commit under test (the GPU box has no .git)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python bench.py ${BENCH_ARGS:-} --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof.json 2>/dev/null
python tools/kstats.py $O/stats/st_kernel_trace.csv 80 > $O/kernel_time_per_step.txt
python tools/timeline.py $O/stats/st_kernel_trace.csv ${BACK:-12} > $O/timeline.txt
python tools/lane_dump.py $O/stats/st_kernel_trace.csv ${BACK:-12} > $O/lanes_one_step.txt
# MFMA utilisation: its own --pmc pass (eager driver), merged into the family table
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES -d $O/pmcm -o m --output-format csv -- python bench.py ${BENCH_ARGS:-} --mode eager --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python tools/mfma_util.py $O/pmcm/m_counter_collection.csv $O/mfma_util.json ${BENCH_ARGS:-} > $O/mfma_util.txt 2>&1
MPMAE_MFMA_JSON=$O/mfma_util.json python tools/families.py $O/stats/st_kernel_trace.csv $O/kernel_families.json ${BENCH_ARGS:-} > $O/kernel_families.txt
cp $O/stats/st_kernel_stats.csv $O/rocprofv3_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmcf -o f --output-format csv -- python bench.py ${BENCH_ARGS:-} --mode eager --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmcw -o w --output-format csv -- python bench.py ${BENCH_ARGS:-} --mode eager --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmcf/f_counter_collection.csv $O/pmcw/w_counter_collection.csv $O/pmc_traffic.json > $O/pmc_traffic.txt
rm -rf $O/stats $O/pmcf $O/pmcw $O/pmcm
if [ -z "$SKIP_PREFIX" ]; then
  python tools/lane_split_time.py > $O/lane_split.txt 2>&1
  python tools/op_table.py > $O/op_table_standalone.txt 2>&1
fi
# the bench line LAST: it reads the family table / PMC file the driver will find committed under profiles/
mkdir -p profiles/r99_tmp && cp $O/kernel_families.json $O/pmc_traffic.json profiles/r99_tmp/ 2>/dev/null
timeout 400 python bench.py ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err
rm -rf profiles/r99_tmp
tail -1 $O/bench.json | cut -c1-1500
head -5 $O/kernel_families.txt; cat $O/timeline.txt | head -4
