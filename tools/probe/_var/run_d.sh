mkdir -p gpurun_out/var
timeout 900 python -m pytest tests -x -q -m gpu -k "descript or v3" 2>&1 | tail -3
WHICH=descript N=32 timeout 300 python tools/bench_disc2d.py 2>&1 | grep -v amdgpu.ids > gpurun_out/var/disc_descript.txt; head -1 gpurun_out/var/disc_descript.txt; grep TOTAL gpurun_out/var/disc_descript.txt
