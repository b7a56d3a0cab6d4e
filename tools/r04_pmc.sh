# Round-4 PMC passes (own runs: --kernel-trace + --pmc only, one counter set per run):
#   (a) the fp32 dominant layer S1 + the layer1 temporal kernels (as tools/r03_pmc_s1.sh: bench.py reads the newest rNN files);
#   (b) the bf16 training kernels on the same two shapes (conv_bench_bf16.py): forward, input gradient, the native weight gradient.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_a_$i $R/gpurun_out/pmc_b_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_a_$i -- python3 $R/tools/conv_bench.py --shapes S1,T1 --kinds fwd,dgrad,wgrad --iters 3 --pre --stats > $R/gpurun_out/pmc_a_$i.log 2>&1 || echo "fp32 set $i failed"
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_a_$i zsv > $R/gpurun_out/r04_pmc_s1_set$i.json 2>/dev/null
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_b_$i -- python3 $R/tools/conv_bench_bf16.py --shapes S1,T1 --kinds fwd,dgrad,wgrad --iters 3 > $R/gpurun_out/pmc_b_$i.log 2>&1 || echo "bf16 set $i failed"
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_b_$i zsv > $R/gpurun_out/r04_pmc_bf16_set$i.json 2>/dev/null
  rm -rf $R/gpurun_out/pmc_a_$i $R/gpurun_out/pmc_b_$i
done
rm -rf $R/gpurun_out/pmc_cal
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_cal -- python3 $R/tools/pmc_calibrate.py > $R/gpurun_out/pmc_cal.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_cal bn_stats > $R/gpurun_out/r04_pmc_calibration.json 2>/dev/null
rm -rf $R/gpurun_out/pmc_cal
