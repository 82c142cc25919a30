cd /root/repo
for o in "" "stem_front=0" "stem_front=0,front_side=0,zero_side=0" "" "stem_front=0" "stem_front=0,front_side=0,zero_side=0"; do MPMAE_ENGINE_OPTS="$o" python tools/fwd_time.py 2>&1 | grep -v amdgpu; done
