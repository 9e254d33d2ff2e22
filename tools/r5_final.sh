# End-of-round batch (one gpurun call): the whole GPU test suite + smoke() on the final code, then the measurements of
# tools/final_measure.sh.  Outputs under gpurun_out/r5f/ (copied into profiles/round5_* afterwards).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export O=gpurun_out/r5f; mkdir -p $O
if [ "${TESTS:-1}" = 1 ]; then
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/pytest_full.log 2>&1; echo "pytest rc $?" >> $O/pytest_full.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
grep -n "passed\|failed\|rc " $O/pytest_full.log | tail -3; tail -2 $O/smoke.log
fi
PARTS="${PARTS:-all}" bash tools/final_measure.sh > $O/final_measure.log 2>&1
tail -70 $O/final_measure.log | cut -c1-330
