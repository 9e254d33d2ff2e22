cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3j}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "discriminator_step_vs_oracle" > $O/tests.log 2>&1; tail -3 $O/tests.log
PARTS=prof2 O=$O bash tools/final_measure.sh
