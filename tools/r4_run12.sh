cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "conv1d or conv2d or encodec or descript or spectral or discrete or v3" > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
timeout 300 python bench.py --config v3 --phase gan --batch 16 --steps 8 --warmup 4 --no-cpu-baseline < /dev/null > $O/bench_v3.log 2>&1; grep "^{" $O/bench_v3.log | cut -c150-290
timeout 300 python bench.py --config discrete --phase gan --batch 32 --steps 8 --warmup 4 --no-cpu-baseline < /dev/null > $O/bench_discrete.log 2>&1; grep "^{" $O/bench_discrete.log | cut -c150-290
WHICH=descript N=32 timeout 300 python tools/bench_disc2d.py < /dev/null > $O/disc_descript.log 2>&1; grep "TOTAL\|fwd+bwd" $O/disc_descript.log | head -8
