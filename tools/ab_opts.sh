# A/B of engine options on the bench workload: OPTS="a=1;b=2,c=3;..." (semicolon-separated option sets)
IFS=';' read -ra SETS <<< "${OPTS:-;ring=20,dz_ring=20}"
for o in "${SETS[@]}"; do
  echo "== $o"; MPMAE_ENGINE_OPTS="$o" python bench.py --steps 40 --warmup 8 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median_hip_events'])"
done
