set -e
R=$GRAFT_REPO_ROOT
RD=${PORL_ROUND:-r03}
mkdir -p $R/gpurun_out/$RD
bash $R/scripts/profile_round.sh > $R/gpurun_out/${RD}_profile.log 2>&1 || { tail -20 $R/gpurun_out/${RD}_profile.log; exit 1; }
cd $R
PORL_PROFILES_OUT=$R/gpurun_out/$RD/profiles_out python scripts/make_profiles.py > $R/gpurun_out/$RD/make_profiles.log 2>&1 || { tail -20 $R/gpurun_out/$RD/make_profiles.log; exit 1; }
python scripts/rocpd_timeline.py $R/gpurun_out/$RD/prof/por_pipelined/t_results.db 100 > $R/gpurun_out/$RD/profiles_out/${RD}_timeline_pipelined.txt 2>&1 || true
rm -rf $R/gpurun_out/$RD/prof/*/    # the databases stay on the box (hundreds of MB); summaries and bench lines travel
ls -la $R/gpurun_out/$RD/profiles_out
tail -25 $R/gpurun_out/$RD/make_profiles.log
