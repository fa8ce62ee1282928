set -u
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "f16f8" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err
python bench.py --k 7 --length 5000 --steps 5 --warmup 2 > gpurun_out/r4/bench_k7_5kb.json 2>/dev/null
python bench.py --rows 200000 --steps 3 --warmup 1 > gpurun_out/r4/bench_200k_rows_1gpu.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_default","bench_k7_5kb","bench_200k_rows_1gpu"):
    try:
        d=json.loads(open('gpurun_out/r4/%s.json'%f).read())
        print(f, d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'count frac', d['roofline_count']['frac'], d['roofline_count']['traffic'], 'verified', d['verified_detail']['worst_error_over_bar'], 'cpu', d.get('cpu_baseline',{}).get('value'), 'x', d.get('speedup_vs_cpu_port'), 'arm', {k:d.get('f16f8_arm',{}).get(k) for k in ('value','ms_per_step','worst_error_over_bar','operand_kind','roofline_frac')})
    except Exception as e: print(f, "no line", e)
PY
