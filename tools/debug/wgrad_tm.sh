cd $GRAFT_REPO_ROOT
for cfg in "0 0" "2 0" "2 768" "2 1536" "3 768" "3 1536" "1 0"; do set -- $cfg
  echo "== TM $1 BLOCKS $2"
  for l in "k3 d3 C192" "k3 d1 C384" "k3 d3 C768" "down k8s4 384" "up k8s4 384"; do
    RH_WGRAD_X6_TM=$1 RH_WGRAD_X6_BLOCKS=$2 ONLY="$l" timeout 60 python tools/bench_layers.py 2>&1 | grep "$l" | cut -c1-100
  done
done
