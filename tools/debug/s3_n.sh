cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3n}; mkdir -p $O
timeout 200 python tools/bench_layers.py < /dev/null > $O/layers.log 2>&1; tail -24 $O/layers.log | cut -c1-100
for rep in 1 2; do
for tn in 2 0; do
  RH_X6_TN=$tn timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-products-leg < /dev/null > $O/bench_tn${tn}_$rep.log 2>&1
  python - <<PY
import json
d=json.loads(open("$O/bench_tn${tn}_$rep.log").read().strip().splitlines()[-1])
print("RH_X6_TN=$tn rep $rep: step", round(d["ms_per_step"],3))
PY
done
done
