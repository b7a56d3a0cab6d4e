# one training-step profile: per-layer standalone table, then the kernel trace of bench.py -> timeline + per-kernel totals
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r03_mid}
python3 $R/tools/conv_bench.py --shapes all --kinds fwd,dgrad,wgrad --iters 10 --pre --stats --add > $R/gpurun_out/${TAG}_conv_layers.txt 2>&1
rm -rf $R/gpurun_out/prof_step
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_step -o x -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_bench_under_rocprof.err
DB=$(find $R/gpurun_out/prof_step -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py $DB 2 multi_tensor_apply > $R/gpurun_out/${TAG}_step_timeline.txt 2>&1
python3 $R/tools/rocpd_stats.py $DB $R/gpurun_out/${TAG}_bench_kernel_stats.csv > /dev/null 2>&1
rm -rf $R/gpurun_out/prof_step
