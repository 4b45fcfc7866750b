#!/bin/bash
# tools/lab/gemm4/build.sh [name] [knob=value ...]  ->  tools/lab/_build/lib_gemm4[_name].so
# The product library with the four-wave GEMM form linked in (gemm.hip rebuilt with -DUSPACE_LAB=1 -DUSPACE_FORM4=1: its 256x256 launches
# consult uspace_lab_gemm_set_big_form; default 1 = the 8-wave template).  With knobs, the K loop text is regenerated into
# tools/lab/_build/var/ first (e.g. `build.sh trace TRACE=2`); without, it is generated there with the default knobs (kloop4.inc is
# generated text and no longer tracked: round 6).
# (the other objects come from the product build: run `make -C uspace_amd/csrc` first)
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../../.." && pwd)
C=$ROOT/uspace_amd/csrc
NAME=$1; shift || true
SUF=${NAME:+_$NAME}
OUT=$ROOT/tools/lab/_build/var_gemm4$SUF
mkdir -p $OUT $ROOT/tools/lab/_build/var
FLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function -DUSPACE_LAB=1 -DUSPACE_FORM4=1"
python3 $HERE/gen_kloop4.py --out=$ROOT/tools/lab/_build/var/kloop4$SUF.inc "$@" > /dev/null
VAR="-DKLOOP4_VARIANT=kloop4$SUF.inc -I$ROOT/tools/lab/_build/var"
/opt/rocm/bin/hipcc $FLAGS -c $C/gemm.hip -o $OUT/gemm.o &
/opt/rocm/bin/hipcc $FLAGS $VAR -c $HERE/gemm4.hip -o $OUT/gemm4.o &
wait
OBJS="$OUT/gemm4.o"
for o in $C/_build/*.o; do
  b=$(basename $o)
  if [ "$b" == "gemm.o" ]; then OBJS="$OBJS $OUT/gemm.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/tools/lab/_build/lib_gemm4$SUF.so
echo built tools/lab/_build/lib_gemm4$SUF.so
