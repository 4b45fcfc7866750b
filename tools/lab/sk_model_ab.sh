#!/bin/bash
# In-model A/B of the GEMM's in-launch K-split tail on ONE box: bench.py --config 3 / 5 (and the batch-16 row count) with the switch
# off and on, alternating, twice.   tools/lab/sk_model_ab.sh [out-file]
OUT=${1:-gpurun_out/r06_sk_model_ab.txt}
mkdir -p $(dirname $OUT); : > $OUT
for rep in 1 2; do
  for cfg in "--config 3" "--config 5" "--config 5 --batch 16"; do
    for sk in 0 1; do
      echo "=== rep $rep: $cfg USPACE_GEMM_SK=$sk" >> $OUT
      USPACE_GEMM_SK=$sk python bench.py $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['unit'], 'ms_per_step', d['ms_per_step'])" >> $OUT 2>&1
    done
  done
done
