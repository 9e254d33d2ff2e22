# session-3: layer table under the K-split variants (finalize launch / in-kernel combine, split targets)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3b}; mkdir -p $O
RH_X6_COMBINE=0 timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_finalize.log 2>&1
timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_combine16.log 2>&1
RH_X6_SPLIT_TARGET=256 timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_combine16_t256.log 2>&1
RH_X6_SPLIT_TARGET=384 timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_combine16_t384.log 2>&1
RH_X6_SPLIT_BELOW=100 timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_combine16_b100.log 2>&1
NIT=10 timeout 200 python tools/check_x6.py < /dev/null > $O/check_x6.log 2>&1; tail -2 $O/check_x6.log
