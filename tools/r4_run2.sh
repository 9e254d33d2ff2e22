cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv2d or encodec or descript or spectral or discrete or v3" > $O/pytest_conv2d.log 2>&1; echo "pytest rc $?" >> $O/pytest_conv2d.log
for w in encodec descript; do WHICH=$w N=32 timeout 300 python tools/bench_disc2d.py < /dev/null > $O/disc_$w.log 2>&1; done
timeout 600 python tools/run_reference_step.py > $O/reference_step.log 2>&1
tail -8 $O/pytest_conv2d.log; grep "TOTAL\|fwd+bwd" $O/disc_encodec.log $O/disc_descript.log; grep "conv2d" $O/disc_encodec.log | head -40; tail -22 $O/reference_step.log | cut -c1-700
