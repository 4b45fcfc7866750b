#!/bin/bash
# tools/lab/build_variant.sh <name> <source.hip of uspace_amd/csrc> "<extra hipcc flags>"  ->  tools/lab/_build/lib_<name>.so
# Defines USPACE_LAB=1 (the lab hooks of gemm.hip: tile-form override, USPACE_CHAIN; older switches are patches under tools/lab/dropped/).
# (the other objects come from the product build: run `make -C uspace_amd/csrc` first)
set -e
NAME=$1; SRC=$2; EXTRA=$3
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/uspace_amd/csrc
mkdir -p $ROOT/tools/lab/_build/var_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function -DUSPACE_LAB=1 $EXTRA -c $C/$SRC -o $ROOT/tools/lab/_build/var_$NAME/${SRC%.hip}.o
OBJS=""
for o in $C/_build/*.o; do
  b=$(basename $o)
  if [ "$b" == "${SRC%.hip}.o" ]; then OBJS="$OBJS $ROOT/tools/lab/_build/var_$NAME/$b"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/tools/lab/_build/lib_$NAME.so
echo built tools/lab/_build/lib_$NAME.so
