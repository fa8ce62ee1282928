set -u
mkdir -p gpurun_out/r4
python -m pytest tests/ -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5
python bench.py -k 8 --rows 20000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4/bench_k8_20k.json 2>/dev/null
python tools/soak.py 600 600 100 100 > gpurun_out/r4/soak_final.log 2>&1; head -c 500 gpurun_out/r4/soak_final.log
