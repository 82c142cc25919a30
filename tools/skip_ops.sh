# Marginal value of op groups in the step (timing experiment, results invalid): SETS="a,b;c;..." -> tools/timing_experiment.py --skip per set
IFS=';' read -ra arr <<< "$SETS"
for o in "${arr[@]}"; do
  echo "== skip: $o"; python tools/timing_experiment.py --skip "$o" -- --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('piece_times'))"
done
