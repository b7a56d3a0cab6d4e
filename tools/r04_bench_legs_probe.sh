#!/bin/bash
# round 4, VERDICT r3 weak #2/#3: where the C3D leg's time outside kernels went, and what holds the host in the timed region.
# One gpurun call, one device: the round-3 behaviour (synced warm-up, unbounded host lead) against the paced loop.
set -x
O=gpurun_out/r04_legs
mkdir -p $O
python bench.py --no-cpu-baseline --no-phases > $O/paced_default.json 2> $O/paced_default.err && \
ZSV_BENCH_PACER_DEPTH=0 ZSV_BENCH_SYNCED_WARMUP=1 python bench.py --no-cpu-baseline --no-phases > $O/r03_behaviour.json 2> $O/r03_behaviour.err && \
python bench.py --network c3d --no-cpu-baseline --no-phases --no-extras > $O/c3d_standalone.json 2> $O/c3d_standalone.err && \
ZSV_BENCH_PACER_DEPTH=0 ZSV_BENCH_SYNCED_WARMUP=1 python bench.py --network c3d --no-cpu-baseline --no-phases --no-extras > $O/c3d_standalone_r03.json 2> $O/c3d_standalone_r03.err && \
python bench.py --steps 100 --no-cpu-baseline --no-phases --no-extras > $O/paced_100.json 2> $O/paced_100.err && \
ZSV_BENCH_PACER_DEPTH=0 python bench.py --steps 100 --no-cpu-baseline --no-phases --no-extras > $O/unpaced_100.json 2> $O/unpaced_100.err && \
ZSV_BENCH_PACER_DEPTH=0 python bench.py --steps 20 --no-cpu-baseline --no-phases --no-extras > $O/unpaced_20.json 2> $O/unpaced_20.err
echo rc=$?
tail -n 3 $O/*.err
