cd $GRAFT_REPO_ROOT
for m in 0 1 2 3; do for k in 0 16 32 48 72; do
  [ $k = 0 ] && [ $m != 0 ] && continue
  echo "== mode $m skew $k"; RH_WGRAD_SKEW=$k RH_WGRAD_SKEW_MODE=$m ONLY="k3 d3 C192" timeout 60 python tools/bench_layers.py 2>&1 | grep "C192\|C768" | cut -c1-100
  RH_WGRAD_SKEW=$k RH_WGRAD_SKEW_MODE=$m ONLY="k3 d3 C768" timeout 60 python tools/bench_layers.py 2>&1 | grep "C768" | cut -c1-100
done; done
