# Marginal value of op groups in the step (timing experiment, results invalid): SETS="a,b;c;..." -> MPMAE_SKIP_OPS per set
IFS=';' read -ra SETS <<< "${SETS:-;head:pix.wgrad;sentinel2.0:pw2.wgrad,sentinel2.0:pw1.wgrad;dw.wgrad;pw.wgrad;.wgrad}"
for o in "${SETS[@]}"; do
  echo "== skip: $o"; MPMAE_SKIP_OPS="$o" python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('piece_times'))"
done
