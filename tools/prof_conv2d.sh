# PMC view of the conv2d kernels on a small Encodec run (separate pass per counter group)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  WHICH=encodec N=2 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc2d_$i -o p -- python $R/tools/bench_disc2d.py > $R/gpurun_out/pmc2d_$i.log 2>&1 < /dev/null
  f=$(find $R/gpurun_out/pmc2d_$i -name "*.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_summary.py $f 2d > $R/gpurun_out/pmc2d_$i.txt 2>&1; fi
  rm -rf $R/gpurun_out/pmc2d_$i
done
cat $R/gpurun_out/pmc2d_*.txt
