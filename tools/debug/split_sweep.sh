cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s; mkdir -p $O
b() { tag=$1; shift; env "$@" timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing < /dev/null > $O/sw_$tag.log 2>&1; echo "$tag $(grep '^{' $O/sw_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), round(d['ms_per_step_median_hip_events'],3))")"; }
b base A=1
b split256 RH_X6_SPLIT_TARGET=256
b split384 RH_X6_SPLIT_TARGET=384
b split768 RH_X6_SPLIT_TARGET=768
b base2 A=1
b wg768 RH_WGRAD_X6_BLOCKS=768
b wg1536 RH_WGRAD_X6_BLOCKS=1536
b base3 A=1
