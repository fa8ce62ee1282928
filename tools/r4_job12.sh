set -u
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_parity.py tests/test_gpu_consumers.py tests/test_gpu_multirank_mock.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -p no:cacheprovider -k "not eight_ranks" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
for rep in 1 2; do
echo "== k=6 self 50000, rep $rep"
python tools/gemm_bench.py --rows 50000 --mode self --rounds 7 --lib seekr_amd/libseekr_hip_prev.so | tail -1
python tools/gemm_bench.py --rows 50000 --mode self --rounds 7 | tail -1
done
echo "== k=7 self 30000"
python tools/gemm_bench.py --rows 30000 --cols 16384 --mode self --tile-operand --rounds 5 --lib seekr_amd/libseekr_hip_prev.so | tail -1
python tools/gemm_bench.py --rows 30000 --cols 16384 --mode self --tile-operand --rounds 5 | tail -1
echo "== k=7 plain stripe 8192 x 100000"
python tools/gemm_bench.py --rows 8192 --rows-b 100000 --cols 16384 --mode plain --tile-operand --rounds 4 --lib seekr_amd/libseekr_hip_prev.so | tail -1
python tools/gemm_bench.py --rows 8192 --rows-b 100000 --cols 16384 --mode plain --tile-operand --rounds 4 | tail -1
echo "== bench k=7"
python bench.py --k 7 --length 5000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4/bench_k7_new.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench_k7_new.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernels_ms_per_step'], d['verified_detail'], d.get('f16f8_arm'))
PY
echo "== rehearsal cfg4 rank 0 (CROSS mode, k=6) and cfg5 rank 0 (k=7 stripes)"
python -m pytest tests/test_gpu_rehearsal.py -x -q -p no:cacheprovider 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
