cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3l}; mkdir -p $O
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-products-leg < /dev/null > $O/bench_n1.log 2>&1
python - <<PY
import json
d=json.loads(open("$O/bench_n1.log").read().strip().splitlines()[-1])
print("step", d["ms_per_step"], "fwd", d["forward_only"])
PY
