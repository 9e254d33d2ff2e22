# headline artefacts only: default bench line, rocprofv3 kernel stats of the same command, layer table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 python bench.py --steps 20 --warmup 5 < /dev/null > gpurun_out/bench_final.log 2>&1
timeout 200 python tools/bench_layers.py < /dev/null > gpurun_out/layers.log 2>&1
bash tools/prof_step.sh > /dev/null 2>&1
grep "^{" gpurun_out/bench_final.log | tail -1 | cut -c1-200; tail -1 gpurun_out/kernel_stats_final.md; grep "unit k1 C96\|TOTAL" gpurun_out/layers.log
