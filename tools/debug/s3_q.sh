# which variant of the discrete (RVQ + spectral discriminator) step records into a hipGraph?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/s3q}; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 200 python -X faulthandler bench.py --config discrete --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --force-graph $PH < /dev/null > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc $(grep '^{' $O/$name.log | tail -1 | cut -c1-0)$(grep -o '"ms_per_step": [0-9.]*' $O/$name.log | tail -1) $(grep -o 'hipGraph[^"]*' $O/$name.log | tail -1 | cut -c1-80)"; }
PH="--phase vae" run vae A=1
PH="--phase gan" run gan_noside RH_BWD_SIDE_STREAM=0
PH="--phase gan" run gan_nolossside RH_LOSS_SIDE_STREAM=0
PH="--phase gan" run gan_nosides RH_BWD_SIDE_STREAM=0 RH_LOSS_SIDE_STREAM=0
PH="--phase gan" run gan_nofm RH_FM_FUSED=0
PH="--phase gan" run gan_torchadam RH_ADAM=0
PH="--phase gan" run gan_nox6_2d RH_CONV2D_X6=0 RH_WGRAD2D_X6=0
PH="--phase gan" run gan_stft_unfused RH_STFT_FUSED=0
