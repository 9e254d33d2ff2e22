cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dispatch.py tests/test_gpu_parity.py -m gpu -q -k "unit or batch32 or residual or reuse or gan_phase or graphed or rvq" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline < /dev/null > $O/bench_n1.log 2>&1
RH_UNIT_FUSED=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline < /dev/null > $O/bench_n1_nofuse.log 2>&1
tail -8 $O/pytest.log; grep "^E  " $O/pytest.log | cut -c1-300 | head -20
python - <<'PY'
import json
for f in ("bench_n1","bench_n1_nofuse"):
    try:
        l=[x for x in open(f"gpurun_out/r3d/{f}.log") if x.startswith("{")][-1]
        d=json.loads(l)
        print(f, d["ms_per_step"], {k:v for k,v in d.get("forward_only",{}).items() if k.startswith("ms")}, d["roofline"]["achieved"], d["roofline"]["launches_per_step"])
    except Exception as e: print(f, "ERR", e); print(open(f"gpurun_out/r3d/{f}.log").read()[-1500:])
PY
