cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -6
for o in "" "stem_front=0"; do MPMAE_ENGINE_OPTS="$o" python tools/fwd_time.py 2>&1 | grep -v amdgpu; done
python tools/prefix_time.py 2>&1 | grep -v amdgpu | head -12
OPTS=";stem_front=0;" bash tools/ab_opts.sh 2>&1 | grep -v amdgpu
