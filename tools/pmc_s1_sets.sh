cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_GATE_EN1"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_s1_$i -- python3 $R/tools/conv_bench.py --shapes S1 --kinds fwd --iters 3 > $R/gpurun_out/pmc_s1_$i.log 2>&1 || echo "set $i failed"
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_s1_$i wino4 2>/dev/null | grep -v dispatches
  rm -rf $R/gpurun_out/pmc_s1_$i
done
