# bf16 training step under rocprofv3: per-kernel totals + the timeline of one step (queues, gaps)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r04_amp}
rm -rf $R/gpurun_out/prof_amp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_amp -o x -- python3 $R/tools/amp_bench.py --steps 6 --optimizer fused > $R/gpurun_out/${TAG}_under_rocprof.txt 2>&1
DB=$(find $R/gpurun_out/prof_amp -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB $R/gpurun_out/${TAG}_kernel_stats.csv > $R/gpurun_out/${TAG}_kernel_stats.txt 2>&1
python3 $R/tools/rocpd_timeline.py $DB 2 adam_multi > $R/gpurun_out/${TAG}_step_timeline.txt 2>&1
rm -rf $R/gpurun_out/prof_amp
