cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
OPTS=";stem_front=0;stem_front=0,front_side=0,zero_side=0;;stem_front=0" bash tools/ab_opts.sh 2>&1 | tee gpurun_out/ab_front.txt
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --profile-names 2>&1 >/dev/null | grep -i "stem\|prep\|mask\|act" | head -20 > gpurun_out/front_names.txt
