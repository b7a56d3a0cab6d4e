# end-of-round evidence (round 4): default bench line, C3D / R3D / fused-Adam / u8 lines, per-layer tables (fp32 + bf16), step timeline +
# per-kernel totals of the fp32 step and of the bf16 training step, HBM bytes per kernel of both steps (own --pmc passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python3 $R/bench.py > $O/r04_bench.json 2> $O/r04_bench.err
python3 $R/bench.py --network c3d --no-cpu-baseline --no-extras > $O/r04_bench_c3d.json 2> /dev/null
python3 $R/bench.py --network r3d_18 --no-cpu-baseline --no-extras > $O/r04_bench_r3d_18.json 2> /dev/null
python3 $R/bench.py --input u8 --no-cpu-baseline --no-extras > $O/r04_bench_input_u8.json 2> /dev/null
python3 $R/bench.py --optimizer fused --no-cpu-baseline --no-extras > $O/r04_bench_fused_adam.json 2> /dev/null
python3 $R/tools/conv_bench.py --shapes all --kinds fwd,dgrad,wgrad --iters 10 --pre --stats > $O/r04_conv_layers_end.txt 2>&1
python3 $R/tools/conv_bench_bf16.py --shapes all > $O/r04_conv_layers_bf16_end.txt 2>&1
rm -rf $O/prof_step
rocprofv3 --kernel-trace --stats -d $O/prof_step -o x -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/r04_bench_under_rocprof.json 2> $O/r04_bench_under_rocprof.err
DB=$(find $O/prof_step -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py $DB 2 multi_tensor_apply > $O/r04_step_timeline.txt 2>&1
python3 $R/tools/rocpd_stats.py $DB $O/r04_bench_kernel_stats.csv > /dev/null 2>&1
rm -rf $O/prof_step
bash $R/tools/r04_amp_profile.sh r04_train_bf16_end
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_step_$c $O/pmc_amp_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_step_$c -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-phases > $O/pmc_step_$c.log 2>&1 || echo "$c pass failed"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_amp_$c -- python3 $R/tools/amp_bench.py --steps 4 > $O/pmc_amp_$c.log 2>&1 || echo "$c bf16 pass failed"
done
# fp32: 2 warm-up + 6 timed + 3 idle-queue steps = 11 whole steps; bf16: 3 warm-up + 4 timed + 1 idle-queue = 8
python3 $R/tools/pmc_step_traffic.py $O/pmc_step_FETCH_SIZE $O/pmc_step_WRITE_SIZE 11 > $O/r04_step_hbm_traffic.txt 2>&1
python3 $R/tools/pmc_step_traffic.py $O/pmc_amp_FETCH_SIZE $O/pmc_amp_WRITE_SIZE 8 > $O/r04_train_bf16_step_hbm_traffic.txt 2>&1
rm -rf $O/pmc_step_FETCH_SIZE $O/pmc_step_WRITE_SIZE $O/pmc_amp_FETCH_SIZE $O/pmc_amp_WRITE_SIZE
head -12 $O/r04_step_hbm_traffic.txt; head -12 $O/r04_train_bf16_step_hbm_traffic.txt
