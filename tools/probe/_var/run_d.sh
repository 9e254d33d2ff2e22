mkdir -p gpurun_out/var
WHICH=v2 N=64 timeout 300 python tools/bench_disc2d.py 2>&1 | grep -v amdgpu.ids > gpurun_out/var/disc_v2_vr.txt; head -1 gpurun_out/var/disc_v2_vr.txt; grep "dgrad.x6\|TOTAL" gpurun_out/var/disc_v2_vr.txt
