cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4w; mkdir -p $O
timeout 200 python tools/ddp_join_timing.py --world 1 --steps 6 > $O/ddp_join_timing_w1.txt 2>&1; tail -6 $O/ddp_join_timing_w1.txt
RH_WN_BATCH=0 timeout 300 python tools/ddp_join_timing.py --steps 4 > $O/ddp_join_timing_nobatch.txt 2>&1; tail -6 $O/ddp_join_timing_nobatch.txt
