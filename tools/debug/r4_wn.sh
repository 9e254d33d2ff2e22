cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4w; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1; tail -8 $O/pytest_full.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
b() { tag=$1; shift; env "$@" timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-products-leg < /dev/null > $O/b_$tag.log 2>&1; echo "$tag $(grep '^{' $O/b_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), round(d['ms_per_step_median_hip_events'],3))")"; }
b new A=1
b old RH_WN_BATCH=0 RH_LOSS_COMBINE=0
