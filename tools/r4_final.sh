cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest_full.log 2>&1; echo "pytest rc $?" >> $O/pytest_full.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
PARTS="${PARTS:-all}" bash tools/final_measure.sh > $O/final_measure.log 2>&1
grep -n "passed\|failed\|rc " $O/pytest_full.log | tail -3; tail -2 $O/smoke.log; tail -60 $O/final_measure.log | cut -c1-330
