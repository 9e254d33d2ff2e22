cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3s}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dispatch.py -m gpu -x -q -k "graphed_discrete" -s > $O/tests.log 2>&1; tail -4 $O/tests.log
grep -n "spread\|passed\|failed" $O/tests.log | tail -5
