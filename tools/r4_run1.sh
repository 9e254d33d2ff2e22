cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python tools/x6_products.py --md $O/x6_products.md > $O/x6_products.log 2>&1
timeout 600 python tools/run_reference_step.py > $O/reference_step.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 < /dev/null > $O/bench_n1.log 2>&1
tail -5 $O/pytest.log; tail -25 $O/x6_products.log; tail -30 $O/reference_step.log; grep "^{" $O/bench_n1.log | cut -c1-400
