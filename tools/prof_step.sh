# rocprofv3 kernel stats of the default bench (writes gpurun_out/kernel_stats_final.md)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-graph > $R/gpurun_out/prof_final.log 2>&1 < /dev/null
f=$(find $R/gpurun_out/prof_final -name "*.db" | head -1)
python $R/tools/prof_summary.py $f > $R/gpurun_out/kernel_stats_final.md 2>&1
rm -rf $R/gpurun_out/prof_final
head -24 $R/gpurun_out/kernel_stats_final.md | cut -c1-150; tail -1 $R/gpurun_out/kernel_stats_final.md
