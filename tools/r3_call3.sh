# round 3, GPU call 3: GPU suite again + bench (default, one-rank RCCL graph, GAN)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 < /dev/null > $O/bench_n1.log 2>&1
RAVE_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline < /dev/null > $O/bench_dist1.log 2>&1
RAVE_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-kernel-timing < /dev/null > $O/bench_dist1_eager.log 2>&1
timeout 300 python bench.py --phase gan --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing < /dev/null > $O/bench_gan.log 2>&1
tail -12 $O/pytest.log
for f in bench_n1 bench_dist1 bench_dist1_eager bench_gan; do echo "== $f"; grep "^{" $O/$f.log | tail -1 | cut -c1-700; grep -i "error\|Traceback" $O/$f.log | head -5; done
python - <<'PY'
import json
for f in ("bench_n1","bench_dist1","bench_dist1_eager"):
    try:
        l=[x for x in open(f"gpurun_out/r3c/{f}.log") if x.startswith("{")][-1]
        d=json.loads(l)
        print(f, d["ms_per_step"], d.get("step_mode"), d.get("step_mode_note"), d.get("forward_only"), d.get("ddp"))
    except Exception as e: print(f, "ERR", e)
PY
