# L2 request volume per kernel of the bench step (one rocprofv3 --pmc pass: TCC_READ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum) -> which
# kernels load the L2 -> CU fabric, the resource the two-lane backward saturates. usage (GPU box): bash tools/l2_requests.sh gpurun_out/<dir>
set -e
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
O=$1; shift; mkdir -p $O
STEPS=6; WARM=2
timeout 900 rocprofv3 --pmc TCC_READ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/l2 -o l2 --output-format csv -- python bench.py "$@" --steps $STEPS --warmup $WARM --no-cpu-baseline > /dev/null 2>&1
python tools/l2_requests.py $O/l2/l2_counter_collection.csv $O/l2/l2_kernel_trace.csv $((STEPS + WARM)) > $O/l2_requests.txt
rm -rf $O/l2
head -50 $O/l2_requests.txt
