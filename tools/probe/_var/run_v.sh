mkdir -p gpurun_out/var
timeout 300 python tools/check_x6.py 2>&1 | grep -v amdgpu.ids | grep "down\|up \|ragged\|msd\|TOTAL\|worst" | cut -c1-150
WHICH=v2 N=64 timeout 300 python tools/bench_disc2d.py 2>&1 | grep -v amdgpu.ids > gpurun_out/var/disc_v2_vr.txt; head -1 gpurun_out/var/disc_v2_vr.txt; grep "TOTAL" gpurun_out/var/disc_v2_vr.txt
