# round 3, GPU call 2: full GPU suite + skew sweep of the layer table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for m in 0 1; do for us in 4 8 12; do
  RH_X6_SKEW_US=$us RH_X6_SKEW_MODE=$m ONLY="unit k" timeout 100 python tools/bench_layers.py < /dev/null > $O/layers_skew_m${m}_us${us}.log 2>&1
done; done
ONLY="unit k" timeout 100 python tools/bench_layers.py < /dev/null > $O/layers_skew_off.log 2>&1
tail -30 $O/pytest.log
for f in $O/layers_skew_*.log; do echo "== $f"; grep -v "^layer\|amdgpu" $f; done
