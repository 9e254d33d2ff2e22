cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3k}; mkdir -p $O
timeout 200 python tools/debug/range_misses.py gan discrete > $O/misses_discrete.log 2>&1; tail -25 $O/misses_discrete.log
timeout 200 python tools/debug/range_misses.py gan v3 > $O/misses_v3.log 2>&1; tail -25 $O/misses_v3.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv2d or disc or encodec or descript or discrete" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 python bench.py --config discrete --phase gan --batch 32 --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing < /dev/null > $O/bench_discrete.log 2>&1; tail -1 $O/bench_discrete.log | cut -c1-330
