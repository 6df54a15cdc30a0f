mkdir -p gpurun_out
rm -rf gpurun_out/prof_trace gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq1 gpurun_out/prof_sq2 gpurun_out/prof_sq3 gpurun_out/prof_longread_trace gpurun_out/prof_site_trace
bash tools/profile_round.sh > gpurun_out/r4_profile_round.log 2>&1; echo "profile_round rc $?"
tail -30 gpurun_out/r4_profile_round.log
