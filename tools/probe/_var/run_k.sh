timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv1d or conv_transpose or golden or weight_norm or first_layer" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pk -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-graph > /dev/null 2>&1 < /dev/null
python $R/tools/prof_summary.py $(find $R/gpurun_out/pk -name "*.db" | head -1) | grep "prep_pack\|total kernel"; rm -rf $R/gpurun_out/pk
