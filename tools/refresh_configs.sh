# bench lines (+ a rocprofv3 kernel summary) of the other BASELINE configs: pix_mod in MX-fp8 and bf16, all_mod tiny 112/16
set -x
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
O=${1:-gpurun_out/configs}; mkdir -p $O
python bench.py --subset pix_mod --dtype fp8 --no-cpu-baseline > $O/bench_pix_mod_fp8.json 2> $O/bench_pix_mod_fp8.err
python bench.py --subset pix_mod --no-cpu-baseline > $O/bench_pix_mod_bf16.json 2> $O/bench_pix_mod_bf16.err
python bench.py --model convnextv2_tiny --img 112 --patch 16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_tiny112_bs256.json 2> $O/bench_tiny112_bs256.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/fp8 -o st --output-format csv -- python bench.py --subset pix_mod --dtype fp8 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python tools/kstats.py $O/fp8/st_kernel_trace.csv 40 > $O/kernel_time_per_step_pix_mod_fp8.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/tiny -o st --output-format csv -- python bench.py --model convnextv2_tiny --img 112 --patch 16 --steps 8 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python tools/kstats.py $O/tiny/st_kernel_trace.csv 40 > $O/kernel_time_per_step_tiny112.txt
rm -f $O/fp8/st_kernel_trace.csv $O/tiny/st_kernel_trace.csv
for f in $O/bench_*.json; do tail -1 $f | cut -c1-330; done
