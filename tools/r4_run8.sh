cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest_full.log 2>&1; echo "pytest rc $?" >> $O/pytest_full.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
PARTS="probes bench others disc products refstep" bash tools/final_measure.sh > $O/final_measure_1.log 2>&1
grep -n "passed\|failed\|rc " $O/pytest_full.log | tail -5; tail -2 $O/smoke.log; tail -40 $O/final_measure_1.log | cut -c1-420
