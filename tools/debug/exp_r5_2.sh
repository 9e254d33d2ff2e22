cd $GRAFT_REPO_ROOT
run() { timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median_hip_events'])"; }
for i in 1 2; do
echo "== default"; run
echo "== BLOCKS=512"; RH_WGRAD_X6_BLOCKS=512 run
echo "== BLOCKS=-1 (resident)"; RH_WGRAD_X6_BLOCKS=-1 run
done
echo "== BLOCKS=384"; RH_WGRAD_X6_BLOCKS=384 run
echo "== BLOCKS=640"; RH_WGRAD_X6_BLOCKS=640 run
