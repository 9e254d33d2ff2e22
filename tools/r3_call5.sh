cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dispatch.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or stft or spectral or graphed or skip_dead" > $O/pytest2.log 2>&1; echo "pytest rc $?" >> $O/pytest2.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline < /dev/null > $O/bench_n1.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/profg -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$O/profg.log 2>&1 < /dev/null)
f=$(find $O/profg -name "*.db" | head -1)
[ -n "$f" ] && python tools/prof_summary.py $f > $O/kernel_stats_step_b32_graph.md 2>&1
rm -rf $O/profg
timeout 200 python tools/prof_ops.py > $O/prof_ops.txt 2>&1
tail -6 $O/pytest.log; grep "^E  " $O/pytest.log | cut -c1-300 | head; tail -4 $O/pytest2.log; grep "^E  " $O/pytest2.log | cut -c1-300 | head
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r3e/bench_n1.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["ms_per_step"], d["forward_only"]["ms"], d["roofline"]["achieved"])
PY
head -60 $O/kernel_stats_step_b32_graph.md; tail -2 $O/kernel_stats_step_b32_graph.md
