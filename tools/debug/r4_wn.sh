cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4w; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_stft_loss.py tests/test_gpu_dispatch.py -q -k "stft or finalize or graph_replay or multiscale" > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
b() { tag=$1; shift; env "$@" timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-products-leg < /dev/null > $O/b_$tag.log 2>&1; echo "$tag $(grep '^{' $O/b_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), round(d['ms_per_step_median_hip_events'],3))")"; }
b one A=1
b per RH_STFT_ONE_FINALIZE=0
b one2 A=1
b per2 RH_STFT_ONE_FINALIZE=0
