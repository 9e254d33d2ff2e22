cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3o}; mkdir -p $O
for b in 0 256 384 768 1024; do
  RH_WGRAD_X6_BLOCKS=$b timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_b$b.log 2>&1
done
RH_WGRAD_X6_TM=3 timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_tm3.log 2>&1
RH_WGRAD_X6_TM=2 timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_tm2.log 2>&1
