# round 3, GPU call 1: probes + GPU test suite + default bench + layer table (outputs under gpurun_out/r3a/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3a; mkdir -p $O
timeout 60 tools/probe/_var/tr_read > $O/tr_read.txt 2>&1
timeout 120 tools/probe/_var/mfma_clock > $O/mfma_clock.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 200 python tools/bench_layers.py < /dev/null > $O/layers.log 2>&1
RH_X6_SWAP=0 timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_noswap.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 < /dev/null > $O/bench_n1.log 2>&1
tail -5 $O/pytest.log; cat $O/mfma_clock.txt; head -20 $O/tr_read.txt; cat $O/layers.log | tail -24; tail -3 $O/layers_noswap.log; grep "^{" $O/bench_n1.log | cut -c1-600
