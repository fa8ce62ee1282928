set -u
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_parity.py -x -q -k "f16f8 or any_alphabet or k8_counts" 2>&1 | tail -8
echo "== strict grid with f16f8"
python tools/strict_parity.py --rows 768 --f16f8 2>&1 | tee gpurun_out/r4/strict_parity_grid_f16f8.log | tail -30
echo "== contraction A/B (normalised rows): f16x3 vs f16f8, 50 000 rows self"
python tools/gemm_bench.py --rows 50000 --mode self --normalised --also f16f8 --rounds 7 2>&1 | tee gpurun_out/r4/gemm_ab_f16f8_k6.log
echo "== k = 7, 30 000 rows self"
python tools/gemm_bench.py --rows 30000 --cols 16384 --mode self --normalised --tile-operand --also f16f8 --rounds 5 2>&1 | tee gpurun_out/r4/gemm_ab_f16f8_k7.log
echo "== bench --precision f16f8"
python bench.py --precision f16f8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4/bench_f16f8.json 2> gpurun_out/r4/bench_f16f8.err; tail -c 500 gpurun_out/r4/bench_f16f8.err; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r4/bench_f16f8.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['kernels_ms_per_step'], d['verified'], d['verified_detail'])
except Exception as e: print("no line", e)
PY
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4/bench_f16x3.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench_f16x3.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline_count']['frac'], d['verified_detail'])
PY
echo "== full-size check with f16f8"
python tools/fullsize_check.py --rows 50000 --precision f16f8 2>&1 | tail -15 | tee gpurun_out/r4/fullsize_f16f8_50k.log
echo "== soak"
python tools/soak.py 700 500 60 60 > gpurun_out/r4/soak_new_rule.log 2>&1; tail -c 2500 gpurun_out/r4/soak_new_rule.log
