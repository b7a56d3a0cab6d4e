# end-of-round evidence: default bench line, C3D / R3D lines, u8 line, per-layer table, step timeline + per-kernel totals
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python3 $R/bench.py > $O/r03_bench.json 2> $O/r03_bench.err
python3 $R/bench.py --network c3d --no-cpu-baseline --no-extras > $O/r03_bench_c3d.json 2> /dev/null
python3 $R/bench.py --network r3d_18 --no-cpu-baseline --no-extras > $O/r03_bench_r3d_18.json 2> /dev/null
python3 $R/bench.py --input u8 --no-cpu-baseline --no-extras > $O/r03_bench_input_u8.json 2> /dev/null
python3 $R/bench.py --optimizer fused --no-cpu-baseline --no-extras > $O/r03_bench_fused_adam.json 2> /dev/null
python3 $R/tools/conv_bench.py --shapes all --kinds fwd,dgrad,wgrad --iters 10 --pre --stats > $O/r03_conv_layers_end.txt 2>&1
rm -rf $O/prof_step
rocprofv3 --kernel-trace --stats -d $O/prof_step -o x -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/r03_bench_under_rocprof.json 2> $O/r03_bench_under_rocprof.err
DB=$(find $O/prof_step -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py $DB 2 multi_tensor_apply > $O/r03_step_timeline.txt 2>&1
python3 $R/tools/rocpd_stats.py $DB $O/r03_bench_kernel_stats.csv > /dev/null 2>&1
rm -rf $O/prof_step
