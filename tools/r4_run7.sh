cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4g; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --config discrete --phase gan --batch 32 --steps 4 --warmup 2 --no-cpu-baseline < /dev/null > $O/disc_$tag.log 2>&1; echo "$tag rc $? $(grep '^{' $O/disc_$tag.log | python -c "import sys,json
l=sys.stdin.readline()
print(json.loads(l)['ms_per_step'], json.loads(l)['step_mode'][:20], str(json.loads(l).get('step_mode_note'))[:200]) if l else print('no json')")"; }
run alloff RH_CONV2D_X6=0 RH_CONV2D_SMALLM=0 RH_WGRAD2D_X6=0
run x6only RH_CONV2D_SMALLM=0 RH_WGRAD2D_X6=0
run smallm RH_CONV2D_X6=0 RH_WGRAD2D_X6=0
run allon
