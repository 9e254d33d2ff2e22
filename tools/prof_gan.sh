# rocprofv3 kernel stats of the v2 GAN-phase bench, eager (writes gpurun_out/kernel_stats_gan.md)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gan -o p -- python $R/bench.py --phase gan --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-graph > $R/gpurun_out/prof_gan.log 2>&1 < /dev/null
f=$(find $R/gpurun_out/prof_gan -name "*.db" | head -1)
python $R/tools/prof_summary.py $f > $R/gpurun_out/kernel_stats_gan.md 2>&1
rm -rf $R/gpurun_out/prof_gan
head -40 $R/gpurun_out/kernel_stats_gan.md | cut -c1-170; tail -1 $R/gpurun_out/kernel_stats_gan.md
