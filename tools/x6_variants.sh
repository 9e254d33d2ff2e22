#!/bin/bash
# Builds one copy of librave_hip.so per timing ablation of conv_x6_kernel (RH_X6_VAR, see conv_x6_kernel.inc) into
# tools/probe/_var/ (git-ignored, travels with gpurun).  Run HERE (hipcc cross-compiles), then on the GPU:
#   for v in 0 1 2 3 4 5 6 7; do RAVE_HIP_LIB=tools/probe/_var/librave_hip_v$v.so ONLY="unit k3 d1 C96" python tools/bench_layers.py; done
set -e
cd "$(dirname "$0")/.."
python -m rave_amd.build >/dev/null
OBJ=rave_amd/_obj
OUT=tools/probe/_var
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
for v in ${VARS:-0 1 2 3 4 5 6 7}; do
  (
    # (the stride-1 shapes only: the layers tools/x6_variants_run.sh times)
    for f in rave_amd/csrc/conv_x6_i1_*.hip; do
      b=$(basename $f .hip)
      /opt/rocm/bin/hipcc $FLAGS -DRH_X6_VAR=$v -x hip -c $f -o $OUT/${b}_v$v.o
    done
    others=$(ls $OBJ/*.o | grep -v "conv_x6_i1_")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/librave_hip_v$v.so $others $OUT/conv_x6_i1_*_v$v.o
    rm -f $OUT/*_v$v.o
  ) &
done
wait
ls -la $OUT
