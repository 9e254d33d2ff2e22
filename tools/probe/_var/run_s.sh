mkdir -p gpurun_out/var
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "first_layer" 2>&1 | tail -3
for t in 768; do echo "== target $t"; RH_SMALLC_WGS=$t WHICH=v2 N=64 timeout 300 python tools/bench_disc2d.py 2>&1 | grep "wgrad.valu" | head -20; done
