#!/bin/bash
# Build an A/B variant of libzsv_hip.so: tools/variant.sh <name> <file.hip> "<extra -D flags>"
# -> build/variants/libzsv_<name>.so (all other objects taken from the regular build).  Load it with ZSV_LIB_PATH.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/zeroshotvideoclassification_amd/csrc
NAME=$1; FILE=$2; FLAGS=$3
OUT=$ROOT/build/variants
mkdir -p $OUT/$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$ROOT/include -I$CSRC -Wall -Wno-unused-function $FLAGS -c $CSRC/$FILE -o $OUT/$NAME/${FILE%.hip}.o
OBJS=""
for o in $CSRC/*.o; do
  b=$(basename $o)
  if [ "$b" == "${FILE%.hip}.o" ]; then OBJS="$OBJS $OUT/$NAME/$b"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libzsv_$NAME.so $OBJS
echo built $OUT/libzsv_$NAME.so
