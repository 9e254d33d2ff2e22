cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "conv2d or encodec or descript or spectral or discrete or v3" > $O/pytest_conv2d.log 2>&1; echo "pytest rc $?" >> $O/pytest_conv2d.log
for w in encodec descript; do WHICH=$w N=32 timeout 300 python tools/bench_disc2d.py < /dev/null > $O/disc_$w.log 2>&1; done
tail -5 $O/pytest_conv2d.log; grep "TOTAL\|fwd+bwd" $O/disc_encodec.log $O/disc_descript.log
