cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3f}; mkdir -p $O
timeout 300 python tools/debug/gloo_capture_recover.py > $O/recover.log 2>&1; grep -v "Warning\|warn\|amdgpu.ids\|hostname" $O/recover.log | cut -c1-220 | tail -40
timeout 600 python -m pytest tests/test_ddp_rccl.py tests/test_ddp_gpu.py -m gpu -x -q > $O/ddp.log 2>&1; tail -3 $O/ddp.log
for i in 1 2 3 4 5 6 7 8; do RAVE_FORCE_DIST=1 timeout 300 python tests/graph_identity_worker.py > $O/gi_$i.log 2>&1; echo "graph identity worker (dist) run $i rc=$?"; done
