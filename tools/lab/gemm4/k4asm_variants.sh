#!/bin/bash
# tools/lab/gemm4/k4asm_variants.sh "name:KNOB=v KNOB=v" ...   -> tools/lab/_build/k4asm_<name> (one K-loop-only harness per schedule variant of
# gen_kloop4.py; summary: `awk -f tools/lab/gemm4/k4asm_summary.awk`)
cd "$(dirname "$0")/.."
mkdir -p _build/var
for v in "$@"; do
  name=${v%%:*}; kn=${v#*:}; [ "$kn" == "$name" ] && kn=""
  ( python3 gemm4/gen_kloop4.py --out=_build/var/kloop4_$name.inc TRACE=1 $kn >/dev/null &&
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -DKLOOP4_INC="\"kloop4_$name.inc\"" -I_build/var \
      gemm4/k4asm_lab.hip -o _build/k4asm_$name -L../../uspace_amd -luspace_hip -Wl,-rpath,'$ORIGIN/../../../uspace_amd' || echo "FAILED $name" ) &
done
wait
