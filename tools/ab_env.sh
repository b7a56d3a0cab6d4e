#!/bin/bash
# A/B of environment toggles on one device: tools/ab_env.sh "<VAR=1 or 'base'> ..." "<conv_bench args>" [reps]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REPS=${3:-2}
for rep in $(seq 1 $REPS); do
  for v in $1; do
    echo "== $v (rep $rep)"
    if [ "$v" == "base" ]; then python $ROOT/tools/conv_bench.py $2 2>&1 | grep -v "^total\|amdgpu.ids"; else env $v python $ROOT/tools/conv_bench.py $2 2>&1 | grep -v "^total\|amdgpu.ids"; fi
  done
done
