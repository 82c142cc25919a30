cd /root/repo
python -m pytest tests/test_hip_kernels_bf16.py -m gpu -x -q -k "stem" 2>&1 | grep -v "^$" | tail -4
for o in "" "stem_front=0"; do MPMAE_ENGINE_OPTS="$o" python tools/fwd_time.py 2>&1 | grep -v amdgpu; done
python tools/prefix_time.py 2>&1 | grep -v amdgpu | head -14
