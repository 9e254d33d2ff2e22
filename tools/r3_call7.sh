cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "stft or spectral or multiscale or golden or graphed or pqmf or splitk or conv1d_fwd" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log; grep "^E  " $O/pytest.log | cut -c1-300 | head
PARTS="layers pmc bench" bash tools/final_measure.sh
