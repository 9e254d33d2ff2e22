cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3r}; mkdir -p $O
for pre in default side none; do echo "== eager pre-steps: $pre"; PRE=$pre timeout 600 python tools/debug/capture_bisect2.py kw_v2 spectral_gen_step 2>&1 | cut -c1-200; done
echo "== default stream, capture_safe=False pre-steps"; PRE=default PRE_SAFE=0 timeout 600 python tools/debug/capture_bisect2.py kw_v2 2>&1 | cut -c1-200
