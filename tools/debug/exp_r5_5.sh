cd $GRAFT_REPO_ROOT
run() { timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median_hip_events'])"; }
for i in 1 2 3; do
echo "== default"; run
echo "== X6_SPLIT_BELOW=200"; RH_X6_SPLIT_BELOW=200 run
echo "== X6_SPLIT_BELOW=260"; RH_X6_SPLIT_BELOW=260 run
done
