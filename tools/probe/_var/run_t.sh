timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "period_major" 2>&1 | tail -12
