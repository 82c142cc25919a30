cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -12
for i in 1 2 3 4; do python -m pytest tests/test_hip_boundary.py -m gpu -x -q --tb=short -k "forward_decoder_sees" 2>&1 | grep -E "assert|Error|passed|failed" | head -8; done
