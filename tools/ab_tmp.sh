timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bf16_step or full_batch_256" 2>&1 | tail -3
for o in "" "ln_fold_defer=0"; do
  echo "== $o"; MPMAE_ENGINE_OPTS="$o" timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median_hip_events'], d.get('piece_times'))"
done
