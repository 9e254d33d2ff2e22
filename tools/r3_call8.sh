cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log; grep "^E  " $O/pytest.log | cut -c1-300 | head
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing < /dev/null > $O/bench_n1.log 2>&1
RH_BWD_SIDE_STREAM=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing < /dev/null > $O/bench_n1_noside.log 2>&1
timeout 300 python bench.py --phase gan --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing < /dev/null > $O/bench_gan.log 2>&1
python - <<'PY'
import json
for f in ("bench_n1","bench_n1_noside","bench_gan"):
    try:
        l=[x for x in open(f"gpurun_out/r3h/{f}.log") if x.startswith("{")][-1]
        d=json.loads(l); print(f, d["ms_per_step"], d["step_mode"])
    except Exception as e: print(f, "ERR", e); print(open(f"gpurun_out/r3h/{f}.log").read()[-800:])
PY
