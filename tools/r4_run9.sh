cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O
timeout 300 python tools/debug/wgrad2d_isolate.py > $O/wgrad2d_isolate.log 2>&1; cat $O/wgrad2d_isolate.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "conv2d or encodec or descript or spectral" > $O/pytest_conv2d.log 2>&1; tail -2 $O/pytest_conv2d.log
WHICH=descript N=32 timeout 300 python tools/bench_disc2d.py < /dev/null > $O/disc_descript.log 2>&1; grep "TOTAL conv2d\|fwd+bwd" $O/disc_descript.log
timeout 300 python bench.py --config v3 --phase gan --batch 16 --steps 8 --warmup 4 --no-cpu-baseline < /dev/null > $O/bench_v3.log 2>&1; grep "^{" $O/bench_v3.log | cut -c1-260
for v in "RH_BWD_SIDE_STREAM=0 --phase gan" "RH_BWD_SIDE_STREAM=1 --phase vae"; do set -- $v; env $1 timeout 200 python bench.py --config discrete $2 $3 --batch 32 --steps 4 --warmup 2 --no-cpu-baseline --force-graph < /dev/null > $O/disc_cap_$3_$1.log 2>&1; echo "capture [$v] rc $? $(grep '^{' $O/disc_cap_$3_$1.log | cut -c150-330)"; done
