mkdir -p gpurun_out/var
timeout 900 python -m pytest tests -x -q -m gpu -k "discrim or period or training_step or golden or hinge" 2>&1 | tail -5
WHICH=v2 N=64 timeout 300 python tools/bench_disc2d.py 2>&1 | grep -v amdgpu.ids > gpurun_out/var/disc_v2_pm.txt; head -3 gpurun_out/var/disc_v2_pm.txt; grep TOTAL gpurun_out/var/disc_v2_pm.txt
