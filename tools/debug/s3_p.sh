cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3p}; mkdir -p $O
timeout 300 python -X faulthandler bench.py --config discrete --phase gan --batch 32 --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing --force-graph < /dev/null > $O/discrete_graph.log 2>&1; echo "rc=$?"; tail -5 $O/discrete_graph.log | cut -c1-400
timeout 300 python -X faulthandler bench.py --config discrete --phase gan --batch 8 --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing --force-graph < /dev/null > $O/discrete_graph_b8.log 2>&1; echo "rc=$?"; tail -5 $O/discrete_graph_b8.log | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_dispatch.py -m gpu -x -q -k "batch32_dispatch or full_width_forward" > $O/tests.log 2>&1; tail -2 $O/tests.log
