# PMC passes over the S1 / S4 Winograd weight gradient: F(4,3) form (8-wave workgroup) against the F(2,3) form
# (ZSV_NO_WGRAD_WINO4=1).  Own runs: --kernel-trace + --pmc only, one counter set per run.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  for arm in on off; do
    rm -rf $R/gpurun_out/pmc_w_$i
    if [ $arm = off ]; then export ZSV_NO_WGRAD_WINO4=1; else unset ZSV_NO_WGRAD_WINO4; fi
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_w_$i -- python3 $R/tools/conv_bench.py --shapes S1,S4 --kinds wgrad --iters 3 > $R/gpurun_out/pmc_w_$i.log 2>&1 || echo "set $i $arm failed"
    python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_w_$i wgrad_wino > $R/gpurun_out/r04_wgrad_f43_${arm}_set$i.json 2>/dev/null
    rm -rf $R/gpurun_out/pmc_w_$i
  done
done
unset ZSV_NO_WGRAD_WINO4
