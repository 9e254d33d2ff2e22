# timeline of one replayed step of the committed tree (tools/prof_timeline.py) -> gpurun_out/r6timeline/timeline.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6timeline; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/tr -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-products-leg > $GRAFT_REPO_ROOT/$O/run.log 2>&1 < /dev/null)
f=$(find $O/tr -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_timeline.py $f $O/timeline.txt > $O/timeline.log 2>&1; rm -rf $O/tr
head -6 $O/timeline.txt
