# session-3: the combine test, the two-rank body over gloo, one bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3d}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dispatch.py -m gpu -x -q -k "combined_inside or graph_replay or graphed" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python -m pytest tests/test_ddp_rccl.py tests/test_ddp_gpu.py -m gpu -x -q > $O/ddp.log 2>&1; tail -3 $O/ddp.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline < /dev/null > $O/bench_n1.log 2>&1; tail -1 $O/bench_n1.log | cut -c1-300
