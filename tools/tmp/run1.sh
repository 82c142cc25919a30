cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 50 --warmup 10 2>/dev/null | tail -1 > gpurun_out/b_front.json
python - <<'PY'
import json; d=json.load(open('gpurun_out/b_front.json')); print(d['ms_per_step'], d.get('ms_per_step_with_input_stage'))
PY
