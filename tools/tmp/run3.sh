cd /root/repo
python -m pytest tests/test_hip_boundary.py -m gpu -x -q -k "forward_decoder_sees" 2>&1 | grep -v "^$" | tail -30
python tools/cpu_enqueue_probe.py 2>&1 | tail -3
