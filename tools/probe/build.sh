# Builds the standalone probes the measurement batch runs (tools/final_measure.sh) into tools/probe/_var/ (git-ignored; the
# binaries travel to the GPU box with the gpurun snapshot).  hipcc cross-compiles: run this in the build container.
set -e
cd "$(dirname "$0")"
mkdir -p _var
for p in mfma_clock tr_read fetch_calib mfma_valu_overlap unaligned_b128 lds_unaligned f16x3 atomic_fanin; do
  [ -f $p.hip ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -o _var/$p $p.hip
done
ls -la _var | grep -v "\.so"
