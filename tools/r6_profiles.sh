# Round-6 evidence with the final library (run from the repo root on the GPU box): rocprofv3 kernel stats + PMC passes
#   gpurun -- 'bash tools/r6_profiles.sh'   then copy gpurun_out/prof_r6_*/r6_*_{kernel_stats.csv,pmc_summary.txt,bench_under_rocprof.json} to profiles/
set -u
bash tools/profile.sh r6_f16x3 f16x3 > gpurun_out/r6_profile_default.log 2>&1
bash tools/profile.sh r6_f16x3_k7 f16x3 --k 7 --length 5000 > gpurun_out/r6_profile_k7.log 2>&1
bash tools/profile.sh r6_f16f8 f16f8 > gpurun_out/r6_profile_f16f8.log 2>&1
T_TRACE=500 T_PMC=400 bash tools/profile.sh r6_f16x3_200k f16x3 --rows 200000 > gpurun_out/r6_profile_200k.log 2>&1
bash tools/profile.sh r6_f16x3_acgtn f16x3 --alphabet ACGTN > gpurun_out/r6_profile_acgtn.log 2>&1
for t in r6_f16x3 r6_f16x3_k7 r6_f16f8 r6_f16x3_200k r6_f16x3_acgtn; do head -3 gpurun_out/prof_$t/${t}_kernel_stats.csv | cut -c1-160; done
