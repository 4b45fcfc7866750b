#!/bin/bash
# rocprofv3 evidence for one round (run on the GPU box from the repo root):  tools/profile_round.sh r02
#   1. --kernel-trace over the default bench command          -> gpurun_out/<tag>_kernel_stats.md
#   2. --pmc SQ_* over three forwards (tools/one_forward.py)  -> gpurun_out/<tag>_pmc_sq.txt
#   3. --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes -> gpurun_out/<tag>_pmc_fetch.txt / _write.txt
# Counter passes carry only --kernel-trace besides --pmc.
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }

# (USPACE_BENCH_EAGER=1: rocprofv3 7.2 segfaults tracing the hipGraph replays the default path uses; same kernels, eager launches)
rm -rf /tmp/p_kt && USPACE_BENCH_EAGER=1 timeout 900 rocprofv3 --kernel-trace -d /tmp/p_kt -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> /tmp/p_kt.err
python $REPO/tools/rocpd_stats.py "$(db /tmp/p_kt)" $OUT/${TAG}_kernel_stats.md > /dev/null || tail -5 /tmp/p_kt.err

rm -rf /tmp/p_sq && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY -d /tmp/p_sq -- python $REPO/tools/one_forward.py > /dev/null 2> /tmp/p_sq.err
python $REPO/tools/rocpd_pmc.py "$(db /tmp/p_sq)" $OUT/${TAG}_pmc_sq.txt > /dev/null || tail -5 /tmp/p_sq.err

rm -rf /tmp/p_f && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -- python $REPO/tools/one_forward.py > /dev/null 2> /tmp/p_f.err
python $REPO/tools/rocpd_pmc.py "$(db /tmp/p_f)" $OUT/${TAG}_pmc_fetch.txt > /dev/null || tail -5 /tmp/p_f.err
rm -rf /tmp/p_w && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -- python $REPO/tools/one_forward.py > /dev/null 2> /tmp/p_w.err
python $REPO/tools/rocpd_pmc.py "$(db /tmp/p_w)" $OUT/${TAG}_pmc_write.txt > /dev/null || tail -5 /tmp/p_w.err
ls -la $OUT | grep ${TAG}_
