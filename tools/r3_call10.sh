cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3j; mkdir -p $O
timeout 600 python tools/debug/precompute_identity.py > $O/ident_two.log 2>&1
grep "differing\|e.g." $O/ident_two.log | cut -c1-90
timeout 600 python tools/debug/precompute_identity.py > $O/ident_two2.log 2>&1
grep "differing\|e.g." $O/ident_two2.log | cut -c1-90
