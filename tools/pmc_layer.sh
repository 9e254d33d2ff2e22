# PMC counters of the conv kernels on single layer shapes of tools/bench_layers.py (ONLY=<substring>), two passes.
#   usage (on the GPU box): bash tools/pmc_layer.sh "<layer substring>" <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L="$1"; T="$2"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
P2="SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  ONLY="$L" timeout 300 rocprofv3 --kernel-trace --pmc $P -d $R/gpurun_out/pmc_$T$i -o p -- python $R/tools/bench_layers.py > $R/gpurun_out/pmc_$T$i.log 2>&1 < /dev/null
  python $R/tools/pmc_summary.py $(find $R/gpurun_out/pmc_$T$i -name "*.db" | head -1) ${KPAT:-conv_x6} > $R/gpurun_out/pmc_$T$i.txt 2>&1
  rm -rf $R/gpurun_out/pmc_$T$i
done
cat $R/gpurun_out/pmc_${T}1.txt $R/gpurun_out/pmc_${T}2.txt
