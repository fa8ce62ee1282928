# Round-6 bench lines kept under profiles/ (run from the repo root on the GPU box, AFTER the matching PMC summaries are in profiles/)
set -u
mkdir -p gpurun_out/r6
python bench.py --steps 20 --warmup 5 > gpurun_out/r6/bench_default.json 2>/dev/null
python bench.py --k 7 --length 5000 --steps 5 --warmup 2 > gpurun_out/r6/bench_k7_5kb.json 2>/dev/null
python bench.py --rows 200000 --steps 3 --warmup 1 > gpurun_out/r6/bench_200k_rows_1gpu.json 2>/dev/null
python bench.py --precision f16f8 --steps 20 --warmup 5 > gpurun_out/r6/bench_f16f8.json 2>/dev/null
python bench.py --alphabet ACGTN --steps 3 --warmup 1 > gpurun_out/r6/bench_acgtn_k6.json 2>/dev/null
python bench.py -k 8 --rows 20000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r6/bench_k8_20k.json 2>/dev/null
for k in 3 4 5; do python bench.py --k $k --steps 20 --warmup 5 --no-target-200k --no-f16f8-arm > gpurun_out/r6/bench_k$k.json 2>/dev/null; done  # k = 3, 4: output bound, priced against HBM
python - <<'PY'
import json
for f in ("bench_default","bench_k7_5kb","bench_200k_rows_1gpu","bench_f16f8","bench_acgtn_k6","bench_k8_20k"):
    try:
        d=json.loads(open('gpurun_out/r6/%s.json'%f).read())
        print(f, d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'count', d['roofline_count']['frac'], d['roofline_count']['traffic'], 'verified', d['verified_detail']['worst_error_over_bar'], 'x', d.get('speedup_vs_cpu_port'), 'arm', (d.get('f16f8_arm') or {}).get('value'))
    except Exception as e: print(f, "no line", e)
PY
