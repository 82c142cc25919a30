for o in "" "dz_ring=3" "dz_ring=4,ring=4" "dz_ring=3,ring=4" "lanes=0"; do
  echo "== $o"; MPMAE_ENGINE_OPTS="$o" python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median_hip_events'])"
done
