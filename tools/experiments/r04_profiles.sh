# all committed measurements of round 4 in one call (GPU box): configs[1], [2], [4] + the decoder's counters
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/profile_round.sh r04 512 c1 > gpurun_out/r04_profile_c1.log 2>&1; echo "c1 rc=$?"; tail -12 gpurun_out/r04_profile_c1.log
bash tools/profile_round.sh r04c2 1024 c2 > gpurun_out/r04_profile_c2.log 2>&1; echo "c2 rc=$?"; tail -12 gpurun_out/r04_profile_c2.log
bash tools/profile_round.sh r04c4 256 c4 > gpurun_out/r04_profile_c4.log 2>&1; echo "c4 rc=$?"; tail -12 gpurun_out/r04_profile_c4.log
bash tools/experiments/pmc_jpeg.sh 256 r04 > gpurun_out/r04_pmc_jpeg.log 2>&1; echo "jpeg rc=$?"; tail -3 gpurun_out/r04_pmc_jpeg.log
