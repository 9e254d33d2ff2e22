cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3m}; mkdir -p $O
timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_default.log 2>&1
RH_X6_TN=1 timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_tn1.log 2>&1
RH_X6_TN=1 RH_X6_SPLIT_TARGET=256 timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_tn1_t256.log 2>&1
RH_X6_TN=1 RH_X6_SPLIT_BELOW=100 timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_tn1_b100.log 2>&1
RH_X6_SPLIT_TARGET=1024 timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_t1024.log 2>&1
