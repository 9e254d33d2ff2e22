cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3g}; mkdir -p $O
timeout 300 python tools/debug/gloo_capture_recover.py > $O/recover.log 2>&1; grep -v "Warning\|warn\|amdgpu.ids\|hostname" $O/recover.log | cut -c1-220 | grep "^[01] " | tail -30
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do RAVE_FORCE_DIST=1 timeout 300 python tests/graph_identity_worker.py > $O/gi_$i.log 2>&1; echo "graph identity worker (dist) run $i rc=$?"; done
