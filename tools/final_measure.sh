# end-of-round measurement batch (run on the GPU box through gpurun); outputs under gpurun_out/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 python bench.py --steps 20 --warmup 5 < /dev/null > gpurun_out/bench_final.log 2>&1
timeout 300 python bench.py --phase gan --steps 8 --warmup 4 --no-cpu-baseline < /dev/null > gpurun_out/bench_gan.log 2>&1
timeout 300 python bench.py --config discrete --phase gan --batch 32 --steps 4 --warmup 4 < /dev/null > gpurun_out/bench_discrete.log 2>&1
timeout 300 python bench.py --config v3 --phase gan --batch 16 --steps 4 --warmup 4 < /dev/null > gpurun_out/bench_v3.log 2>&1
timeout 300 python tools/bench_layers.py < /dev/null > gpurun_out/layers.log 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof_final.log 2>&1 < /dev/null)
f=$(find gpurun_out/prof_final -name "*.db" | head -1)
[ -n "$f" ] && python tools/prof_summary.py $f > gpurun_out/kernel_stats_final.md 2>&1
rm -rf gpurun_out/prof_final
tail -c 600 gpurun_out/bench_final.log; echo; tail -c 300 gpurun_out/bench_gan.log; echo; tail -c 300 gpurun_out/bench_discrete.log; echo; tail -c 300 gpurun_out/bench_v3.log; echo; tail -2 gpurun_out/layers.log; head -12 gpurun_out/kernel_stats_final.md
# opt-in mode and the data-parallel path exercised with one rank (RCCL init, buckets, hooks)
timeout 300 python bench.py --phase gan --skip-dead-grads --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing < /dev/null > gpurun_out/bench_gan_skip.log 2>&1
RAVE_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing < /dev/null > gpurun_out/bench_dist1.log 2>&1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing < /dev/null > gpurun_out/bench_torchrun1.log 2>&1
for w in v2 encodec descript; do N=32; [ $w = v2 ] && N=64; WHICH=$w N=$N timeout 300 python tools/bench_disc2d.py < /dev/null > gpurun_out/disc_$w.log 2>&1; done
tail -c 400 gpurun_out/bench_gan_skip.log; echo; tail -c 500 gpurun_out/bench_dist1.log; echo; tail -c 300 gpurun_out/bench_torchrun1.log; echo
grep "TOTAL\|fwd+bwd" gpurun_out/disc_v2.log gpurun_out/disc_encodec.log gpurun_out/disc_descript.log
