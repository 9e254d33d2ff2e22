# SQ counters (matrix-core busy cycles, issue stalls, LDS conflicts) of the x6 kernels on two layer shapes of the v2 model
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
KPAT=conv_x6 bash tools/pmc_layer.sh "unit k3 d3 C768" c768 > gpurun_out/pmcl_c768_conv.txt 2>&1
KPAT=wgrad_x6 bash tools/pmc_layer.sh "unit k3 d3 C768" w768 > gpurun_out/pmcl_c768_wgrad.txt 2>&1
KPAT=conv_x6 bash tools/pmc_layer.sh "unit k3 d1 C96" c96 > gpurun_out/pmcl_c96_conv.txt 2>&1
KPAT=wgrad_x6 bash tools/pmc_layer.sh "unit k3 d1 C96" w96 > gpurun_out/pmcl_c96_wgrad.txt 2>&1
tail -30 gpurun_out/pmcl_c768_conv.txt
