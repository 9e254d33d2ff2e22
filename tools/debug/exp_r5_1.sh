cd $GRAFT_REPO_ROOT; O=gpurun_out/r5; mkdir -p $O
echo "== stft occ3 (product)"; timeout 100 python tools/bench_stft_loss.py 2>&1 | grep "sum over"
echo "== stft occ2 variant"; RAVE_HIP_LIB=$PWD/rave_amd/_var/librave_hip_occ2.so timeout 100 python tools/bench_stft_loss.py 2>&1 | grep "sum over"
for i in 1 2; do
echo "== bench default"; timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median_hip_events'])"
echo "== bench RH_GRAPH_PRIORITY=1"; RH_GRAPH_PRIORITY=1 timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median_hip_events'])"
done
echo "== bench occ2 lib"; RAVE_HIP_LIB=$PWD/rave_amd/_var/librave_hip_occ2.so timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median_hip_events'])"
for t in 512 768 1536; do echo "== bench RH_WGRAD_X6_BLOCKS=$t"; RH_WGRAD_X6_BLOCKS=$t timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median_hip_events'])"; done
