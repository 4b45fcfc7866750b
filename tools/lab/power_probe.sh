#!/bin/bash
# Lab: board power and shader clock (rocm-smi) sampled once a second while tools/one_forward.py runs U-ViT-L forwards back to back.
cd $GRAFT_REPO_ROOT
rocm-smi --showmaxpower 2>/dev/null | grep -E "Max"
python tools/one_forward.py --reps ${1:-4000} > /dev/null 2>&1 &
PID=$!
for i in $(seq 1 240); do
  kill -0 $PID 2>/dev/null || break
  P=$(rocm-smi --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$")
  C=$(rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | grep -oE "\([0-9]+Mhz\)")
  T=$(rocm-smi --showtemp 2>/dev/null | grep -E "junction" | grep -oE "[0-9.]+$")
  echo "t=$i power=$P sclk=$C junction=$T"
  sleep 1
done
wait $PID 2>/dev/null
