set -u
mkdir -p gpurun_out/r4
python tools/fullsize_check.py --rows 200000 --precision f16f8 2>&1 | tail -8 | tee gpurun_out/r4/fullsize_f16f8_200k.log
SEEKR_PRECISION=f16f8 python tools/soak.py 900 900 5 5 > gpurun_out/r4/soak_f16f8_long.log 2>&1; head -c 600 gpurun_out/r4/soak_f16f8_long.log; echo
python tools/soak.py 700 700 100 100 > gpurun_out/r4/soak_default_3.log 2>&1; head -c 600 gpurun_out/r4/soak_default_3.log
