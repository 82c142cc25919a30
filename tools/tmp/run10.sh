cd /root/repo
python -m pytest tests/test_hip_kernels_bf16.py tests/test_hip_parity.py -m gpu -x -q -k "stem or golden or oracle or fallback or drivers" 2>&1 | grep -v "^$" | tail -6
for o in "" "stem_front=0" "stem_front=0,front_side=0,zero_side=0"; do MPMAE_ENGINE_OPTS="$o" python tools/fwd_time.py 2>&1 | grep -v amdgpu; done
python tools/prefix_time.py 2>&1 | grep -v amdgpu | head -14
