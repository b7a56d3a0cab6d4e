cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_c3d
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_c3d -o x -- python3 $R/tools/amp_bench.py --network c3d --steps 6 --optimizer fused > $R/gpurun_out/r04_c3d_bf16_under_rocprof.txt 2>&1
DB=$(find $R/gpurun_out/prof_c3d -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py $DB 2 adam_multi > $R/gpurun_out/r04_c3d_bf16_step_timeline.txt 2>&1
rm -rf $R/gpurun_out/prof_c3d
grep "^#" $R/gpurun_out/r04_c3d_bf16_step_timeline.txt | head -32
