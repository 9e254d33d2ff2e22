cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "conv2d or encodec or descript or spectral or discrete or v3" > $O/pytest_conv2d.log 2>&1; echo "pytest rc $?" >> $O/pytest_conv2d.log
for w in encodec descript; do WHICH=$w N=32 timeout 300 python tools/bench_disc2d.py < /dev/null > $O/disc_$w.log 2>&1; done
timeout 300 python bench.py --config discrete --phase gan --batch 32 --steps 8 --warmup 4 --no-cpu-baseline < /dev/null > $O/bench_discrete.log 2>&1
tail -8 $O/pytest_conv2d.log; grep "TOTAL\|fwd+bwd" $O/disc_encodec.log $O/disc_descript.log; grep "conv2d_wgrad" $O/disc_encodec.log | head -12;  grep "^{" $O/bench_discrete.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_mode'], d.get('step_mode_note'))"
