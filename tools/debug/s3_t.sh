cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3t}; mkdir -p $O
timeout 300 python tools/debug/two_stream_forward.py > $O/two_stream.log 2>&1; tail -3 $O/two_stream.log
