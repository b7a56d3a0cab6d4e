# PMC passes (own runs, --kernel-trace + --pmc only) over the shipped temporal kernels of layer1: T1 forward (BatchNorm-folded form),
# dgrad, wgrad (folded form) -- matrix-pipe busy / clock, then HBM bytes.  Run on the GPU box: bash tools/r03_pmc_temporal.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--shapes T1 --kinds fwd,dgrad,wgrad --iters 3 --pre"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_t1_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_t1_$i -- python3 $R/tools/conv_bench.py $ARGS > $R/gpurun_out/pmc_t1_$i.log 2>&1 || echo "set $i failed"
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_t1_$i zsv > $R/gpurun_out/r03_pmc_t1_set$i.json 2>/dev/null
  rm -rf $R/gpurun_out/pmc_t1_$i
done
