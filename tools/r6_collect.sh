# copies the outputs of tools/r6_final.sh (gpurun_out/r6final) into profiles/round6_* (run in the build container)
set -e
cd "$(dirname "$0")/.."
O=${1:-gpurun_out/r6final}
grep "^{" $O/bench_n1.log | tail -1 > profiles/round6_bench_n1.json
( for f in bench_n1_eager bench_dist1 bench_gan bench_gan_skip bench_discrete bench_v3; do grep "^{" $O/$f.log | tail -1; done ) > profiles/round6_bench_other_configs.jsonl
cp $O/dispatches_per_step.txt profiles/round6_dispatches_per_step.txt
cp $O/kernel_stats_per_replayed_step.md profiles/round6_kernel_stats_per_replayed_step.md
cp $O/kernel_stats_step_b32.md profiles/round6_kernel_stats_step_b32.md
cp $O/kernel_stats_step_b32_graph.md profiles/round6_kernel_stats_step_b32_graph.md
cp $O/layers.log profiles/round6_layer_table_b32.txt
cp $O/check_x6.log profiles/round6_x6_vs_f32_layers.txt
cp $O/pqmf.log profiles/round6_pqmf_kernels.txt
cp $O/stft_loss.log profiles/round6_stft_loss_kernels.txt
cp $O/pmc_traffic.json profiles/round6_pmc_traffic.json
for c in FETCH_SIZE WRITE_SIZE; do cp $O/pmc_$c.txt profiles/round6_pmc_$c.txt; cp $O/pmc_calib_$c.txt profiles/round6_pmc_calib_$c.txt; done
( echo "full GPU suite on the final round-6 tree (tools/r6_final.sh: one gpurun call, then the measurements of profiles/round6_* on the same box):"; tail -1 $O/gpu_tests.log; tail -1 $O/smoke.log ) > profiles/round6_gpu_tests.txt
sha256sum rave_amd/librave_hip.so | cut -c1-16; grep -o '"librave_hip_sha256": "[0-9a-f]\{16\}' profiles/round6_pmc_traffic.json
