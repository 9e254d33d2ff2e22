cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_v3 -o p -- python $GRAFT_REPO_ROOT/bench.py --config v3 --phase gan --batch 16 --steps 8 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_v3.log 2>&1 < /dev/null)
f=$(find $O/prof_v3 -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f > $O/kernel_stats_v3_gan_b16.md 2>&1; rm -rf $O/prof_v3
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_d -o p -- python $GRAFT_REPO_ROOT/bench.py --config discrete --phase gan --batch 32 --steps 8 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_discrete.log 2>&1 < /dev/null)
f=$(find $O/prof_d -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f > $O/kernel_stats_discrete_gan_b32.md 2>&1; rm -rf $O/prof_d
head -45 $O/kernel_stats_v3_gan_b16.md | cut -c1-200; head -45 $O/kernel_stats_discrete_gan_b32.md | cut -c1-200
