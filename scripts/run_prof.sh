set -e
R=$GRAFT_REPO_ROOT
bash $R/scripts/profile_r02.sh > $R/gpurun_out/r02_profile.log 2>&1 || { tail -20 $R/gpurun_out/r02_profile.log; exit 1; }
cd $R
PORL_PROFILES_OUT=$R/gpurun_out/r02/profiles_out python scripts/make_profiles_r02.py > $R/gpurun_out/r02/make_profiles.log 2>&1 || { tail -20 $R/gpurun_out/r02/make_profiles.log; exit 1; }
python scripts/rocpd_timeline.py $R/gpurun_out/r02/prof/por_pipelined/t_results.db 100 > $R/gpurun_out/r02/profiles_out/r02_timeline_pipelined.txt 2>&1 || true
rm -rf $R/gpurun_out/r02/prof/*/    # the databases stay on the box (hundreds of MB); summaries and bench lines travel
ls -la $R/gpurun_out/r02/profiles_out
tail -25 $R/gpurun_out/r02/make_profiles.log
