# session-3 final batch: the full GPU test suite, then tools/final_measure.sh on the same box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/r6final}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
PARTS="pmc" O=$O bash tools/final_measure.sh; cp $O/pmc_traffic.json profiles/round6_pmc_traffic.json  # (stamp first: the bench line below then carries the traffic)
PARTS="bench others layers prof count" O=$O bash tools/final_measure.sh
