mkdir -p gpurun_out/var
for all in 0 1; do for l in "C96" "stem" "out k7"; do
RH_WGRAD_X6_ALL=$all ONLY="$l" timeout 120 python tools/bench_layers.py 2>&1 | grep -v "^layer\|^TOTAL\|amdgpu.ids" | sed "s/^/all=$all /"
done; done | tee gpurun_out/var/wgrad_all.txt
