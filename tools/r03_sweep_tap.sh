# sweep of the direct kernel's tile (ZSV_CONV_CFG) and K parts (ZSV_TAP_KS) on the small-voxel layers it still runs
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_tap_sweep.txt
: > $OUT
for shape in T9 T10 T8 S8 T5 S5; do
  echo "== $shape default" >> $OUT
  python3 $R/tools/conv_bench.py --shapes $shape --kinds fwd,dgrad --iters 10 2>&1 | grep -v "total\|amdgpu" >> $OUT
  for cfg in 0 1 2 3; do
    for ks in 1 2 3 4 6 8 12; do
      echo "== $shape cfg $cfg ks $ks" >> $OUT
      ZSV_CONV_CFG=$cfg ZSV_TAP_KS=$ks python3 $R/tools/conv_bench.py --shapes $shape --kinds fwd --iters 10 2>&1 | grep -v "total\|amdgpu" >> $OUT
    done
  done
done
