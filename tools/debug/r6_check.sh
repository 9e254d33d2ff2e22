# the full GPU suite + smoke + the default bench invocation, one box (a last check of the committed tree)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6check
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6check/gpu_tests.log 2>&1; tail -2 gpurun_out/r6check/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6check/smoke.log 2>&1; tail -1 gpurun_out/r6check/smoke.log
( time timeout 600 python bench.py ) > gpurun_out/r6check/bench_default.log 2>&1; grep "^{" gpurun_out/r6check/bench_default.log | cut -c1-200; grep real gpurun_out/r6check/bench_default.log
