#!/bin/bash
# GPU side of tools/x6_variants.sh: forward / data-gradient timings of a few layers under every ablation build
mkdir -p gpurun_out/var
for v in ${VARS:-0 1 2 3 4 5 6 7}; do
  for l in "unit k3 d1 C96" "unit k1 C96" "unit k3 d3 C192" "unit k1 C192" "unit k3 d1 C384" "unit k3 d3 C768"; do
    RAVE_HIP_LIB=$PWD/tools/probe/_var/librave_hip_v$v.so ONLY="$l" timeout 120 python tools/bench_layers.py 2>&1 | grep -v "^layer\|^TOTAL" | sed "s/^/v$v /"
  done
done | tee gpurun_out/var/variants.txt
