cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -40
for i in 1 2 3; do python -m pytest tests/test_hip_boundary.py -m gpu -x -q 2>&1 | tail -3; done
