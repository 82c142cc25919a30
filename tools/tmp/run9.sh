cd /root/repo
python tools/prefix_time.py 2>&1 | grep -v amdgpu > gpurun_out/prefix_default.txt
MPMAE_ENGINE_OPTS="stem_front=0,front_side=0,zero_side=0" python tools/prefix_time.py 2>&1 | grep -v amdgpu > gpurun_out/prefix_old.txt
