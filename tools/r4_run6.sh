cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv2d" > $O/pytest_conv2d.log 2>&1; tail -2 $O/pytest_conv2d.log
WHICH=encodec N=32 timeout 300 python tools/bench_disc2d.py < /dev/null > $O/disc_encodec.log 2>&1; grep "TOTAL\|fwd+bwd" $O/disc_encodec.log; grep "^conv2d" $O/disc_encodec.log | awk '$2<=4 || $3<=4' | head -12
timeout 300 python -X faulthandler bench.py --config discrete --phase gan --batch 32 --steps 6 --warmup 2 --no-cpu-baseline < /dev/null > $O/bench_discrete.log 2>&1
echo "discrete rc $?"; tail -40 $O/bench_discrete.log | cut -c1-260
