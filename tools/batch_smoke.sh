# The recorded step at other per-GPU batch sizes / dtypes / models (runs, finite loss): usage (GPU box) bash tools/batch_smoke.sh [out file]
O=${1:-gpurun_out/batch_smoke.txt}
: > $O
run() {
  r=$(timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms  final_loss', d['config']['final_loss'])
except Exception as e:
    print('FAILED')")
  echo "$*: $r" | tee -a $O
}
for b in 1 2 3 17 64 100 255 512 1024; do run --batch $b; done
for b in 1 7 256; do run --batch $b --dtype f32; done
for b in 1 5 64 256; do run --batch $b --model convnextv2_tiny --img 112 --patch 16; done
for b in 3 256 700; do run --batch $b --subset pix_mod; done
for b in 3 256; do run --batch $b --subset pix_mod --dtype fp8; done
run --batch 32 --model convnextv2_base --img 112 --patch 16
run --batch 64 --model convnextv2_nano --img 56 --patch 8
