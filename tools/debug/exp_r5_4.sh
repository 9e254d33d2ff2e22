cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_dispatch.py -q -p no:cacheprovider -k "staged_once_per_position or one_launch_equals" 2>&1 | tail -8
for pl in 1 0; do echo "== layer table RH_WGRAD_X6_PLANES=$pl"; RH_WGRAD_X6_PLANES=$pl timeout 200 python tools/bench_layers.py 2>/dev/null | awk '{print $1,$2,$3,$4, $(NF-1), $NF}' | tail -24; done
run() { timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median_hip_events'])"; }
for i in 1 2; do echo "== bench planes on"; run; echo "== bench planes off"; RH_WGRAD_X6_PLANES=0 run; done
