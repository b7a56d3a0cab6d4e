cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_step
rocprofv3 --kernel-trace --stats -d $O/prof_step -o x -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_bench.json 2> $O/prof_bench.err
DB=$(find $O/prof_step -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB $O/prof_kernel_stats.csv > /dev/null 2>&1
rm -rf $O/prof_step
