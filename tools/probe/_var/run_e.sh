mkdir -p gpurun_out/var
timeout 300 python tools/check_x6.py 2>&1 | grep -v amdgpu.ids | tail -12
timeout 300 python tools/bench_layers.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/var/layers_epi.txt
