cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3i}; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -8 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
