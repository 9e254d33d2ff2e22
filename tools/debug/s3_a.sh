# session-3 quick check: conv parity tests, x6 check, layer table, bench (run on the GPU box through gpurun)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3a}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dispatch.py tests/test_gpu_parity.py -m gpu -x -q -k "conv or x6 or v2 or unit or graph or batch32" > $O/tests.log 2>&1; tail -5 $O/tests.log
NIT=10 timeout 200 python tools/check_x6.py < /dev/null > $O/check_x6.log 2>&1; tail -3 $O/check_x6.log
timeout 200 python tools/bench_layers.py < /dev/null > $O/layers.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline < /dev/null > $O/bench_n1.log 2>&1; tail -1 $O/bench_n1.log | cut -c1-400
