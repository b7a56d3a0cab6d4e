# per-kernel breakdown of the layer3/4 and strided convolutions (pack / main / reduce) -- run on the GPU box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SH=${1:-S5,T5,S6,T6,S7,T7,S8,T8,S9,T9,S10,T10}
OUT=${2:-r03_layer34_breakdown}
rm -rf $R/gpurun_out/probe_l34
rocprofv3 --kernel-trace -d $R/gpurun_out/probe_l34 -o x -- python3 $R/tools/conv_bench.py --markers --shapes $SH --kinds fwd,dgrad,wgrad --iters 5 > $R/gpurun_out/$OUT.stdout 2> $R/gpurun_out/$OUT.err
DB=$(find $R/gpurun_out/probe_l34 -name "*.db" | head -1)
python3 $R/tools/kernel_breakdown.py $DB $R/gpurun_out/$OUT.stdout > $R/gpurun_out/$OUT.txt 2>&1
rm -rf $R/gpurun_out/probe_l34
