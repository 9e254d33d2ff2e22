# session-3: v2 step with the K-split combined in the launch (RH_X6_COMBINE = largest slice count combined) against the finalize launches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3c}; mkdir -p $O
for rep in 1 2; do
for c in 0 16 8 4; do
  RH_X6_COMBINE=$c timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-products-leg < /dev/null > $O/bench_c${c}_$rep.log 2>&1
  python - <<PY
import json
d=json.loads(open("$O/bench_c${c}_$rep.log").read().strip().splitlines()[-1])
print("combine<=$c rep $rep: step", round(d["ms_per_step"],3), "ms  fwd", round(d["forward_only"]["ms"],3), "ms")
PY
done
done
timeout 600 python -m pytest tests/test_ddp_rccl.py -m gpu -x -q > $O/ddp.log 2>&1; tail -3 $O/ddp.log
