cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
RH_BWD_SIDE_STREAM=1 timeout 600 python -m pytest tests -m gpu -q -k "graphed or training_step_golden or two_ranks or skip_dead or hot_path_backward" > $O/pytest_side.log 2>&1; echo "pytest rc $?" >> $O/pytest_side.log
timeout 200 python tools/bench_pqmf.py > $O/pqmf.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing < /dev/null > $O/bench_n1.log 2>&1
RH_BWD_SIDE_STREAM=1 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing < /dev/null > $O/bench_n1_side.log 2>&1
RH_BWD_SIDE_STREAM=1 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-graph < /dev/null > $O/bench_n1_side_eager.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-graph < /dev/null > $O/bench_n1_eager.log 2>&1
tail -6 $O/pytest.log; grep "^E  " $O/pytest.log | cut -c1-300 | head; tail -4 $O/pytest_side.log; grep "^E  " $O/pytest_side.log | cut -c1-300 | head
cat $O/pqmf.log | grep -v amdgpu
python - <<'PY'
import json
for f in ("bench_n1","bench_n1_side","bench_n1_side_eager","bench_n1_eager"):
    try:
        l=[x for x in open(f"gpurun_out/r3f/{f}.log") if x.startswith("{")][-1]
        d=json.loads(l); print(f, d["ms_per_step"], d["step_mode"])
    except Exception as e: print(f, "ERR", e); print(open(f"gpurun_out/r3f/{f}.log").read()[-800:])
PY
