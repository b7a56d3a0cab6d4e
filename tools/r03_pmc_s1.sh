# PMC passes (own runs: --kernel-trace + --pmc only) over the dominant layer S1 (forward with statistics, dgrad, wgrad) and the
# layer1 temporal kernels (forward in the BatchNorm-folded form, dgrad, wgrad): matrix-pipe busy / clock, HBM read / write bytes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--shapes S1,T1 --kinds fwd,dgrad,wgrad --iters 3 --pre --stats"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_s1_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_s1_$i -- python3 $R/tools/conv_bench.py $ARGS > $R/gpurun_out/pmc_s1_$i.log 2>&1 || echo "set $i failed"
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_s1_$i zsv > $R/gpurun_out/r03_pmc_s1_set$i.json 2>/dev/null
  rm -rf $R/gpurun_out/pmc_s1_$i
done
# FETCH_SIZE calibration on a known byte count (bn_stats reads the 635.8 MB layer1 mid tensor once)
rm -rf $R/gpurun_out/pmc_cal
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_cal -- python3 $R/tools/pmc_calibrate.py > $R/gpurun_out/pmc_cal.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_cal bn_stats > $R/gpurun_out/r03_pmc_calibration.json 2>/dev/null
rm -rf $R/gpurun_out/pmc_cal
