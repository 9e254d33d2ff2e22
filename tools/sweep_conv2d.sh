for sf in 15360 9600 6400; do for wl in 81920 38912 24576; do
echo "== STAGE_FLOATS=$sf WGRAD_LDS=$wl"; RH_CONV2D_STAGE_FLOATS=$sf RH_WGRAD2D_LDS_BYTES=$wl WHICH=encodec N=4 timeout 200 python tools/bench_disc2d.py 2>&1 < /dev/null | grep "TOTAL\|fwd+bwd"; done; done
