#!/bin/bash
# Interleaved A/B of the round-5 library (lab build of the previous commit) against the current one on the U-ViT GEMM shapes at the row
# counts of every BASELINE configuration and a few in between:  tools/lab/sk_ab.sh [out-file]   (run on the GPU box)
OUT=${1:-gpurun_out/r06_sk_ab.txt}
mkdir -p $(dirname $OUT)
: > $OUT
for D in 1024 512; do
  for M in 21376 8224 16448 4112 12336 24672 2056; do
    echo "=== M=$M D=$D (A = round 5, B = round 6)" >> $OUT
    timeout 300 tools/lab/_build/gemm_ab tools/lab/_build/lib_r05.so tools/lab/_build/lib_r06.so $M 5 10 $D >> $OUT 2>&1
  done
done
