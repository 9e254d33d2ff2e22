cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median_hip_events'], d['step_mode'][:12])"; }
for i in 1 2 3; do
echo "== default"; run
echo "== RH_LOSS_SIDE_STREAM=1"; RH_LOSS_SIDE_STREAM=1 run
done
echo "== eager default"; run --no-graph
echo "== eager RH_LOSS_SIDE_STREAM=1"; RH_LOSS_SIDE_STREAM=1 run --no-graph
RH_LOSS_SIDE_STREAM=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dispatch.py -q -p no:cacheprovider -k "graphed or training_step_golden" 2>&1 | tail -3
