# Same-session A/B of two BUILDS of the library on the bench workload (GPU box): LIBS="path_a;path_b;..." (each is copied over
# mmearth-train_amd/libmpmae_hip.so in turn; the last one stays). Builds go under mmearth-train_amd/build/ (git-ignored, shipped by gpurun).
IFS=';' read -ra SETS <<< "${LIBS:-mmearth-train_amd/build/lib_base.so;mmearth-train_amd/build/lib_new.so}"
for r in ${ROUNDS:-1 2}; do
for l in "${SETS[@]}"; do
  cp "$l" mmearth-train_amd/libmpmae_hip.so
  echo "== $l"; python bench.py --steps 40 --warmup 8 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median_hip_events'], d.get('piece_times'))"
done; done
