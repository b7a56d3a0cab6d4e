# HBM bytes per kernel over a profiled run of bench.py (two own --pmc passes: FETCH_SIZE, WRITE_SIZE; --kernel-trace only next to them)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_step_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_step_$c -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-phases > $R/gpurun_out/pmc_step_$c.log 2>&1 || echo "$c pass failed"
done
# 2 warm-up + 6 timed steps = 8 whole training steps in the run (no phase legs)
python3 $R/tools/pmc_step_traffic.py $R/gpurun_out/pmc_step_FETCH_SIZE $R/gpurun_out/pmc_step_WRITE_SIZE 8 > $R/gpurun_out/r03_step_hbm_traffic_raw.txt 2>&1
rm -rf $R/gpurun_out/pmc_step_FETCH_SIZE $R/gpurun_out/pmc_step_WRITE_SIZE
head -50 $R/gpurun_out/r03_step_hbm_traffic_raw.txt
