#!/bin/bash
# A/B timing of library variants on one device: tools/ab.sh "<variants>" "<conv_bench args>" [reps]
# variant "base" = the in-tree library; others = build/variants/libzsv_<name>.so
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REPS=${3:-2}
for rep in $(seq 1 $REPS); do
  for v in $1; do
    if [ "$v" == "base" ]; then unset ZSV_LIB_PATH; else export ZSV_LIB_PATH=$ROOT/build/variants/libzsv_$v.so; fi
    echo "== $v (rep $rep)"
    python $ROOT/tools/conv_bench.py $2 2>&1 | grep -v "^total"
  done
done
