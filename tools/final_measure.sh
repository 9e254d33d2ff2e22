# end-of-round measurement batch (run on the GPU box through gpurun); outputs under $O (default gpurun_out/r6f/; copied into profiles/round6_*)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/r6f}; mkdir -p $O
PARTS=${PARTS:-all}
has() { [ "$PARTS" = all ] || echo " $PARTS " | grep -q " $1 "; }
if has probes; then      # (binaries: bash tools/probe/build.sh in the build container)
  timeout 120 tools/probe/_var/mfma_clock > $O/probe_mfma_clock.txt 2>&1
  timeout 60 tools/probe/_var/tr_read > $O/probe_tr_read.txt 2>&1
  timeout 60 tools/probe/_var/lds_unaligned > $O/probe_lds_unaligned.txt 2>&1
  timeout 60 tools/probe/_var/f16x3 > $O/probe_f16x3.txt 2>&1
  timeout 60 tools/probe/_var/atomic_fanin > $O/probe_atomic_fanin.txt 2>&1
fi
if has bf16; then        # the comparison build (three bf16 pieces, six products: rave_amd/_var/librave_hip_bf16.so) on the same box
  RAVE_HIP_LIB=$PWD/rave_amd/_var/librave_hip_bf16.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-products-leg < /dev/null > $O/bench_n1_bf16x6.log 2>&1
  RAVE_HIP_LIB=$PWD/rave_amd/_var/librave_hip_bf16.so timeout 200 python tools/bench_layers.py < /dev/null > $O/layers_bf16x6.log 2>&1
fi
if has refstep; then     # the unmodified reference training_step on the drop-ins (needs tools/run_reference_step.sh stage)
  [ -d oracle/_ref/reference/rave ] && timeout 600 python tools/run_reference_step.py > $O/reference_step.log 2>&1
fi
if has bench; then
  timeout 400 python bench.py --steps 20 --warmup 5 < /dev/null > $O/bench_n1.log 2>&1
  timeout 200 python bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-kernel-timing < /dev/null > $O/bench_n1_eager.log 2>&1
  RAVE_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing < /dev/null > $O/bench_dist1.log 2>&1
fi
if has others; then
  timeout 300 python bench.py --phase gan --steps 8 --warmup 4 --no-cpu-baseline < /dev/null > $O/bench_gan.log 2>&1
  timeout 300 python bench.py --phase gan --skip-dead-grads --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing < /dev/null > $O/bench_gan_skip.log 2>&1
  timeout 300 python bench.py --config discrete --phase gan --batch 32 --steps 8 --warmup 4 --no-cpu-baseline < /dev/null > $O/bench_discrete.log 2>&1
  timeout 300 python bench.py --config v3 --phase gan --batch 16 --steps 8 --warmup 4 --no-cpu-baseline < /dev/null > $O/bench_v3.log 2>&1
fi
if has layers; then
  timeout 200 python tools/bench_layers.py < /dev/null > $O/layers.log 2>&1
  NIT=10 timeout 200 python tools/check_x6.py < /dev/null > $O/check_x6.log 2>&1
  timeout 100 python tools/bench_pqmf.py < /dev/null > $O/pqmf.log 2>&1
  timeout 100 python tools/bench_stft_loss.py < /dev/null > $O/stft_loss.log 2>&1
  timeout 100 python tools/debug/split_chain.py < /dev/null > $O/split_chain.log 2>&1
fi
if has prof; then
  (cd /tmp && RH_BWD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-graph > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 < /dev/null)
  f=$(find $O/prof -name "*.db" | head -1)
  [ -n "$f" ] && python tools/prof_summary.py $f > $O/kernel_stats_step_b32.md 2>&1
  rm -rf $O/prof
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/profg -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$O/profg.log 2>&1 < /dev/null)
  f=$(find $O/profg -name "*.db" | head -1)
  [ -n "$f" ] && python tools/prof_summary.py $f > $O/kernel_stats_step_b32_graph.md 2>&1
  rm -rf $O/profg
fi
if has count; then    # dispatches per REPLAYED step: two graph runs that differ only in the number of timed steps (everything else --
                      # warm-up, capture, set-up -- cancels in the difference)
  for n in 20 60; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/cnt$n -o p -- python $GRAFT_REPO_ROOT/bench.py --steps $n --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$O/cnt$n.log 2>&1 < /dev/null)
    f=$(find $O/cnt$n -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f > $O/cnt$n.md 2>&1
  done
  python tools/prof_diff.py $(find $O/cnt20 -name "*.db" | head -1) $(find $O/cnt60 -name "*.db" | head -1) 40 $O/kernel_stats_per_replayed_step.md > /dev/null 2>&1
  rm -rf $O/cnt20 $O/cnt60
  python - <<PY > $O/dispatches_per_step.txt
import re
def tot(p):
    t = open(p).read().strip().splitlines()[-1]
    m = re.search(r"total kernel time ([0-9.]+) ms over (\d+) dispatches", t)
    return float(m.group(1)), int(m.group(2))
(a_ms, a_n), (b_ms, b_n) = tot("$O/cnt20.md"), tot("$O/cnt60.md")
print(f"graph replay, v2 VAE-phase step, batch 32 x 65536: {(b_n - a_n) / 40:.1f} dispatches and {(b_ms - a_ms) / 40:.3f} ms of kernel time per replayed step")
print(f"(rocprofv3 --kernel-trace of bench.py --steps 60 minus --steps 20: {b_n} - {a_n} dispatches, {b_ms:.2f} - {a_ms:.2f} ms)")
PY
  cat $O/dispatches_per_step.txt
fi
if has prof2; then     # rocprofv3 kernel stats of the discrete (eager) and v3 (graph) GAN-phase steps
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_v3 -o p -- python $GRAFT_REPO_ROOT/bench.py --config v3 --phase gan --batch 16 --steps 8 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_v3.log 2>&1 < /dev/null)
  f=$(find $O/prof_v3 -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f > $O/kernel_stats_v3_gan_b16.md 2>&1; rm -rf $O/prof_v3
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_d -o p -- python $GRAFT_REPO_ROOT/bench.py --config discrete --phase gan --batch 32 --steps 8 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_discrete.log 2>&1 < /dev/null)
  f=$(find $O/prof_d -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f > $O/kernel_stats_discrete_gan_b32.md 2>&1; rm -rf $O/prof_d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_g -o p -- python $GRAFT_REPO_ROOT/bench.py --phase gan --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$O/prof_gan.log 2>&1 < /dev/null)
  f=$(find $O/prof_g -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f > $O/kernel_stats_gan_step.md 2>&1; rm -rf $O/prof_g
fi
if has pmc; then
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && RH_BWD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/$O/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1 < /dev/null)
    python tools/pmc_summary.py $(find $O/pmc_$c -name "*.db" | head -1) > $O/pmc_$c.txt 2>&1
    (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/$O/cal_$c -o p -- $GRAFT_REPO_ROOT/tools/probe/_var/fetch_calib > $GRAFT_REPO_ROOT/$O/cal_$c.log 2>&1 < /dev/null)
    python tools/pmc_summary.py $(find $O/cal_$c -name "*.db" | head -1) > $O/pmc_calib_$c.txt 2>&1
  done
  python tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc_WRITE_SIZE -name "*.db" | head -1) $O/pmc_traffic.json \
         $(find $O/cal_FETCH_SIZE -name "*.db" | head -1) $(find $O/cal_WRITE_SIZE -name "*.db" | head -1) > $O/pmc_traffic.log 2>&1
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/cal_FETCH_SIZE $O/cal_WRITE_SIZE
fi
if has disc; then
  for w in v2 encodec descript; do N=32; [ $w = v2 ] && N=64; WHICH=$w N=$N timeout 300 python tools/bench_disc2d.py < /dev/null > $O/disc_$w.log 2>&1; done
fi
for f in bench_n1 bench_n1_bf16x6 bench_n1_eager bench_dist1 bench_gan bench_gan_skip bench_discrete bench_v3; do [ -f $O/$f.log ] && { echo "== $f"; grep "^{" $O/$f.log | tail -1 | cut -c1-330; }; done
[ -f $O/layers.log ] && tail -2 $O/layers.log; [ -f $O/check_x6.log ] && tail -1 $O/check_x6.log; [ -f $O/pqmf.log ] && grep "kernel level\|module" $O/pqmf.log; [ -f $O/stft_loss.log ] && grep "sum over" $O/stft_loss.log
[ -f $O/kernel_stats_step_b32_graph.md ] && { head -12 $O/kernel_stats_step_b32_graph.md; tail -1 $O/kernel_stats_step_b32_graph.md; }
[ -f $O/pmc_traffic.log ] && cat $O/pmc_traffic.log | head -80
[ -f $O/disc_v2.log ] && grep "TOTAL\|fwd+bwd" $O/disc_v2.log $O/disc_encodec.log $O/disc_descript.log
true
