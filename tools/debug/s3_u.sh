cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3u}; mkdir -p $O
timeout 100 python tools/bench_pqmf.py < /dev/null > $O/pqmf.log 2>&1; grep "kernel level\|module" $O/pqmf.log
timeout 100 python tools/bench_pqmf.py < /dev/null > $O/pqmf2.log 2>&1; grep "kernel level" $O/pqmf2.log
timeout 600 python -m pytest tests -m gpu -x -q -k "pqmf" > $O/tests.log 2>&1; tail -2 $O/tests.log
