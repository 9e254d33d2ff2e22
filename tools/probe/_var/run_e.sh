mkdir -p gpurun_out/var
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv1d or wgrad or weight or unit or full_width" 2>&1 | tail -4
timeout 300 python tools/bench_layers.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/var/layers_pp.txt
