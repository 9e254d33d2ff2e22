# the full GPU suite + the headline bench line (traffic stamped from profiles/round6_pmc_traffic.json) + the other configs, one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/r6final2}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
PARTS="bench others" O=$O bash tools/final_measure.sh
