timeout 300 python -m pytest tests -x -q -m gpu -k "pqmf" 2>&1 | tail -2
timeout 100 python tools/bench_pqmf.py 2>&1 | grep fold
