mkdir -p gpurun_out/var
for v in 0 6 7 8; do
  for l in "unit k3 d3 C192" "unit k1 C192" "unit k3 d1 C384" "unit k3 d3 C768" "down k8s4 192->384" "up k8s4 384->192"; do
    RAVE_HIP_LIB=$PWD/tools/probe/_var/librave_hip_w$v.so ONLY="$l" timeout 120 python tools/bench_layers.py 2>&1 | grep -v "^layer\|^TOTAL\|amdgpu.ids" | sed "s/^/w$v /"
  done
done | tee gpurun_out/var/wvariants.txt
