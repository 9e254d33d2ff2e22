# session-3: the 2-D path on the f16 pieces -- parity tests, then the other configs' bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3h}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv2d or disc or encodec or descript or spectral or v3 or discrete or rvq" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 300 python bench.py --config discrete --phase gan --batch 32 --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing < /dev/null > $O/bench_discrete.log 2>&1; tail -1 $O/bench_discrete.log | cut -c1-330
timeout 300 python bench.py --config v3 --phase gan --batch 16 --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing < /dev/null > $O/bench_v3.log 2>&1; tail -1 $O/bench_v3.log | cut -c1-330
timeout 300 python bench.py --phase gan --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing < /dev/null > $O/bench_gan.log 2>&1; tail -1 $O/bench_gan.log | cut -c1-330
