cd /root/repo
for i in 1 2 3 4 5 6; do python -m pytest tests/test_hip_boundary.py -m gpu -x -q --tb=short -k "chained or forward_decoder_sees or pieces" 2>&1 | grep -E "assert|Error|passed|failed" | head -8; done
echo "--- stem_front=0"
for i in 1 2 3 4 5 6; do MPMAE_ENGINE_OPTS="stem_front=0" python -m pytest tests/test_hip_boundary.py -m gpu -x -q --tb=short -k "chained or forward_decoder_sees or pieces" 2>&1 | grep -E "assert|Error|passed|failed" | head -8; done
