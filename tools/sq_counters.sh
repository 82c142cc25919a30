# SQ counter passes over an eager bench run (separate rocprofv3 --pmc passes, never combined with other trace domains)
R=$PWD; O=$R/${1:-gpurun_out/sq}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python $R/bench.py ${BENCH_ARGS:-} --mode eager --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/pmc_summary.py $O/p$i/p_counter_collection.csv 60 > $O/sq_counters_$i.txt 2>&1
  rm -rf $O/p$i
done
