cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
timeout 60 tools/probe/_var/lds_unaligned > $O/probe_lds_unaligned.txt 2>&1
timeout 300 python tools/debug/conv2d_f32_dgrad_bug.py > $O/dgrad_bug.log 2>&1
timeout 300 python bench.py --config discrete --phase gan --batch 32 --steps 8 --warmup 4 --no-cpu-baseline < /dev/null > $O/bench_discrete.log 2>&1
timeout 300 python bench.py --config v3 --phase gan --batch 16 --steps 8 --warmup 4 --no-cpu-baseline < /dev/null > $O/bench_v3.log 2>&1
cat $O/probe_lds_unaligned.txt; cat $O/dgrad_bug.log | cut -c1-300; for f in bench_discrete bench_v3; do grep "^{" $O/$f.log | cut -c1-700; tail -3 $O/$f.log | cut -c1-300; done
