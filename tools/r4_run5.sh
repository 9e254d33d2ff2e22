cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "conv2d or encodec or descript or spectral or discrete or v3" > $O/pytest_conv2d.log 2>&1; echo "pytest rc $?" >> $O/pytest_conv2d.log
for w in encodec descript; do WHICH=$w N=32 timeout 300 python tools/bench_disc2d.py < /dev/null > $O/disc_$w.log 2>&1; done
(cd /tmp && WHICH=encodec N=32 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o p -- python $GRAFT_REPO_ROOT/tools/bench_disc2d.py > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 < /dev/null)
f=$(find $O/prof -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f > $O/kernel_stats_encodec_pass.md 2>&1; rm -rf $O/prof
timeout 300 python -X faulthandler bench.py --config discrete --phase gan --batch 32 --steps 6 --warmup 2 --no-cpu-baseline < /dev/null > $O/bench_discrete.log 2>&1
tail -5 $O/pytest_conv2d.log; grep "TOTAL\|fwd+bwd" $O/disc_encodec.log $O/disc_descript.log; head -40 $O/kernel_stats_encodec_pass.md; tail -30 $O/bench_discrete.log | cut -c1-300
