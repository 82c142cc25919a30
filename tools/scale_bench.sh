# One command for the first multi-GPU measurement: bench lines at N = 1, 2, 4, 8 (whatever the node exposes) into
# profiles/<round>/scale_bench.jsonl, each preceded by the CPU rehearsal of the same rank count (bench.py --dry-run: spawn, rendezvous, bucket
# plan, exchange - no kernels), so that a failure names its layer.   usage: bash tools/scale_bench.sh [profiles/r05]
O=${1:-profiles/r05}; mkdir -p $O; : > $O/scale_bench.jsonl
G=$(python -c "import torch; print(torch.cuda.device_count())")
for n in 1 2 4 8; do
  [ "$n" -le "$G" ] || { echo "{\"n_gpus\": $n, \"skipped\": \"node exposes $G GPU(s)\"}" >> $O/scale_bench.jsonl; continue; }
  if [ "$n" -gt 1 ]; then python bench.py --gpus $n --dry-run --steps 2 --warmup 1 2>/dev/null | tail -1 >> $O/scale_bench.jsonl; fi
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python bench.py --gpus $n --steps 30 --warmup 10 --no-cpu-baseline 2>$O/scale_bench_n$n.err | tail -1 >> $O/scale_bench.jsonl
done
cat $O/scale_bench.jsonl | cut -c1-300
