# two separate PMC passes (never combined with trace domains other than --kernel-trace) -> gpurun_out/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmc_$c.log 2>&1 < /dev/null
  python $R/tools/pmc_summary.py $(find $R/gpurun_out/pmc_$c -name "*.db" | head -1) > $R/gpurun_out/pmc_$c.txt 2>&1
done
python $R/tools/pmc_traffic.py $(find $R/gpurun_out/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $R/gpurun_out/pmc_WRITE_SIZE -name "*.db" | head -1) $R/gpurun_out/pmc_traffic.json
rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
