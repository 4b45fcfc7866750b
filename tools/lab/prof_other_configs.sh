# rocprofv3 --kernel-trace summaries of the BASELINE configurations besides the headline (run on the GPU box from the repo root):
#   tools/lab/prof_other_configs.sh r05   ->  gpurun_out/r05_kernel_stats_config{3,4,5}.md   (10 forwards each; config 1: USPACE profile_round.sh handles 2)
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "3 L_t 64" "4 S_t 64" "5 L_u 32" "1 S_u 4"; do set -- $c; rm -rf /tmp/p_$2_$3; timeout 600 rocprofv3 --kernel-trace -d /tmp/p_$2_$3 -- python $R/tools/one_forward.py --model $2 --batch $3 --reps 10 > /dev/null 2> /tmp/err_$2.txt; python $R/tools/rocpd_stats.py "$(find /tmp/p_$2_$3 -name '*.db' | head -1)" $R/gpurun_out/${TAG}_kernel_stats_config$1.md > /dev/null || tail -3 /tmp/err_$2.txt; done
ls -la $R/gpurun_out/${TAG}_kernel_stats_config*
