for rk in 32; do for blk in 1024 2048 4096; do for mc in 4 8; do
  echo "== rk=$rk blocks=$blk minchunks=$mc"
  for l in "C96" "stem" "out k7"; do
    RH_WGRAD_RK=$rk RH_WGRAD_BLOCKS=$blk RH_WGRAD_MINCHUNKS=$mc ONLY="$l" timeout 120 python tools/bench_layers.py 2>&1 | grep -v "^layer\|^TOTAL\|amdgpu.ids" | awk '{printf "%s %s %s %s  wgrad %s us\n",$1,$2,$3,$4,$(NF-1)}'
  done
done; done; done
