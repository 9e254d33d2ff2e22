mkdir -p gpurun_out/r2
timeout 400 python bench.py --steps 20 --warmup 5 < /dev/null > gpurun_out/r2/bench_n1.log 2>&1
timeout 300 python bench.py --phase gan --steps 8 --warmup 4 --no-cpu-baseline < /dev/null > gpurun_out/r2/bench_gan.log 2>&1
grep "^{" gpurun_out/r2/bench_n1.log | cut -c1-300; grep "^{" gpurun_out/r2/bench_gan.log | cut -c1-300
