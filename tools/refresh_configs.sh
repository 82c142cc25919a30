# The other BASELINE configs (4: all_mod tiny 112/16; 5: pix_mod in bf16 and with the MX-fp8 decoder) measured like the headline: kernel
# trace -> per-step table + in-step families, PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes), then the bench line that reads them.
# usage: MPMAE_COMMIT=<sha> bash tools/refresh_configs.sh gpurun_out/<dir>
set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
O=${1:-gpurun_out/configs}; mkdir -p $O profiles/r99_tmp
one() {   # name, bench args...
  n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/st_$n -o st --output-format csv -- python bench.py "$@" --steps 8 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python tools/kstats.py $O/st_$n/st_kernel_trace.csv 50 > $O/kernel_time_per_step_$n.txt
  python tools/families.py $O/st_$n/st_kernel_trace.csv $O/kernel_families_$n.json "$@" > $O/kernel_families_$n.txt
  BENCH_ARGS="$*" timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf_$n -o f --output-format csv -- python bench.py "$@" --mode eager --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  BENCH_ARGS="$*" timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw_$n -o w --output-format csv -- python bench.py "$@" --mode eager --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  BENCH_ARGS="$*" python tools/pmc_traffic.py $O/pf_$n/f_counter_collection.csv $O/pw_$n/w_counter_collection.csv $O/pmc_traffic_$n.json > $O/pmc_traffic_$n.txt
  rm -rf $O/st_$n $O/pf_$n $O/pw_$n
  cp $O/kernel_families_$n.json $O/pmc_traffic_$n.json profiles/r99_tmp/
  timeout 400 python bench.py "$@" --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
  tail -1 $O/bench_$n.json | cut -c1-400
}
one tiny112_bs256 --model convnextv2_tiny --img 112 --patch 16 --steps 20 --warmup 5
one pix_mod_bf16 --subset pix_mod
one pix_mod_fp8 --subset pix_mod --dtype fp8
rm -rf profiles/r99_tmp
