cd $GRAFT_REPO_ROOT
run() { timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median_hip_events'])"; }
echo "== default (wgrad slices 512)"; run
echo "== X6_SPLIT_TARGET=256"; RH_X6_SPLIT_TARGET=256 run
echo "== X6_SPLIT_TARGET=384"; RH_X6_SPLIT_TARGET=384 run
echo "== X6_SPLIT_BELOW=200"; RH_X6_SPLIT_BELOW=200 run
echo "== X6_SPLIT_BELOW=130"; RH_X6_SPLIT_BELOW=130 run
echo "== default"; run
echo "== RH_WN_BATCH_MAX=16"; RH_WN_BATCH_MAX=16 run
