set -u
bash tools/profile.sh r4_f16x3 f16x3 > gpurun_out/r4_profile_default.log 2>&1
bash tools/profile.sh r4_f16x3_k7 f16x3 --k 7 --length 5000 > gpurun_out/r4_profile_k7.log 2>&1
head -6 gpurun_out/prof_r4_f16x3/r4_f16x3_kernel_stats.csv | cut -c1-200
head -4 gpurun_out/prof_r4_f16x3_k7/r4_f16x3_k7_kernel_stats.csv | cut -c1-200
