# Round-5 evidence with the final library (run from the repo root on the GPU box): rocprofv3 kernel stats + PMC passes
#   gpurun -- 'bash tools/r5_profiles.sh'   then copy gpurun_out/prof_r5_*/r5_*_{kernel_stats.csv,pmc_summary.txt,bench_under_rocprof.json} to profiles/
set -u
bash tools/profile.sh r5_f16x3 f16x3 > gpurun_out/r5_profile_default.log 2>&1
bash tools/profile.sh r5_f16x3_k7 f16x3 --k 7 --length 5000 > gpurun_out/r5_profile_k7.log 2>&1
bash tools/profile.sh r5_f16f8 f16f8 > gpurun_out/r5_profile_f16f8.log 2>&1
T_TRACE=500 T_PMC=400 bash tools/profile.sh r5_f16x3_200k f16x3 --rows 200000 > gpurun_out/r5_profile_200k.log 2>&1
bash tools/profile.sh r5_f16x3_acgtn f16x3 --alphabet ACGTN > gpurun_out/r5_profile_acgtn.log 2>&1
for t in r5_f16x3 r5_f16x3_k7 r5_f16f8 r5_f16x3_200k r5_f16x3_acgtn; do head -3 gpurun_out/prof_$t/${t}_kernel_stats.csv | cut -c1-160; done
