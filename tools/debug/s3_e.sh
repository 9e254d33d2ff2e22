cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3e}; mkdir -p $O
timeout 300 python tools/debug/gloo_capture_recover.py > $O/recover.log 2>&1; grep -v "Warning\|warn" $O/recover.log | tail -60
timeout 900 python -m pytest tests/test_gpu_dispatch.py -m gpu -x -q -k "combined_inside" > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2 3; do RAVE_FORCE_DIST=1 timeout 300 python tests/graph_identity_worker.py > $O/gi_$i.log 2>&1; echo "graph identity worker (dist) run $i rc=$?"; done
