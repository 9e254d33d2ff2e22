cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py "$@" --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median_hip_events'])"; }
for i in 1 2; do
echo "== gan default(200)"; run --phase gan --steps 12 --warmup 4
echo "== gan BELOW=384"; RH_X6_SPLIT_BELOW=384 run --phase gan --steps 12 --warmup 4
done
echo "== v3 default(200)"; run --config v3 --phase gan --batch 16 --steps 8 --warmup 4
echo "== v3 BELOW=384"; RH_X6_SPLIT_BELOW=384 run --config v3 --phase gan --batch 16 --steps 8 --warmup 4
echo "== discrete default(200)"; run --config discrete --phase gan --batch 32 --steps 8 --warmup 4
echo "== discrete BELOW=384"; RH_X6_SPLIT_BELOW=384 run --config discrete --phase gan --batch 32 --steps 8 --warmup 4
