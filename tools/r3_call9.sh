cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
K="graphed or v3_generator or full_width or hot_path_backward or training_step_golden"
timeout 900 python -m pytest tests -m gpu -q -k "$K" > $O/pytest_a.log 2>&1; echo "pytest rc $?" >> $O/pytest_a.log
RH_STFT_PRECOMPUTE=0 timeout 900 python -m pytest tests -m gpu -q -k "graphed" > $O/pytest_b.log 2>&1; echo "pytest rc $?" >> $O/pytest_b.log
RH_BWD_SIDE_STREAM=0 timeout 900 python -m pytest tests -m gpu -q -k "graphed" > $O/pytest_c.log 2>&1; echo "pytest rc $?" >> $O/pytest_c.log
for f in a b c; do echo "== $f"; tail -4 $O/pytest_$f.log; grep "^FAILED\|^E   .*Error" $O/pytest_$f.log | cut -c1-200 | head; done
