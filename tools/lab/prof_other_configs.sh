cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "S_t 64" "L_u 32"; do set -- $c; rm -rf /tmp/p_$1_$2; timeout 600 rocprofv3 --kernel-trace -d /tmp/p_$1_$2 -- python $R/tools/one_forward.py --model $1 --batch $2 --reps 10 > /dev/null 2> /tmp/err_$1.txt; python $R/tools/rocpd_stats.py "$(find /tmp/p_$1_$2 -name '*.db' | head -1)" $R/gpurun_out/kstats_$1_$2.md > /dev/null || tail -3 /tmp/err_$1.txt; done
ls -la $R/gpurun_out/kstats_*
